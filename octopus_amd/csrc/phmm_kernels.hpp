// HIP kernels of the pair-HMM haplotype-likelihood path for gfx950 (MI355X).
//
// Reference behaviour reproduced (paths relative to /root/reference/src):
//   k_classify   HaplotypeLikelihoodModel::max_score candidate logic (core/models/haplotype_likelihood_model.cpp:187-259),
//                hmm::detail::try_naive_evaluate (core/models/pairhmm/pair_hmm.hpp:278-319),
//                window / flank tests of simd_evaluate_helper (pair_hmm.hpp:123-137,731-739)
//   k_dp         simd::PairHMM::align_helper (core/models/pairhmm/simd_pair_hmm.hpp:240-324) incl. update_traceback (:147-163)
//   k_walk       set_alignments (:165-231) fused with calculate_flank_score_helper (:347-430) and the flank discount of
//                simd_evaluate_helper (pair_hmm.hpp:754-764)
//   k_epilogue   mapping-quality mixture of HaplotypeLikelihoodModel::evaluate (haplotype_likelihood_model.cpp:285-303) and the
//                per-template sum (:313-320)
//
// DP layout: one band diagonal per lane, B lanes per task row, 64/B rows per wave, and TWO tasks per row packed in
// the low/high int16 halves of every VGPR (v_pk_add_u16 / v_pk_min_i16 wrap and compare exactly like the reference's
// _mm_add_epi16 / _mm_min_epi16 lanes). Lane i of a row owns cells (t = k-i, x = k+i) and (t, x+1) at iteration k, so both
// the read index and the haplotype index advance by one per iteration and every lane reads its own operands straight
// from LDS; only the D and I states move between lanes (one DPP shift each).
#pragma once
#include "phmm_device.hpp"

namespace octphmm {

constexpr uint32_t INF2 = 0x78007800u;   // infinity_ = SHRT_MAX - 0x7FF in both halves (simd_pair_hmm.hpp:55-56)
constexpr uint32_t NUL2 = 0x80008000u;   // null_score_ = SHRT_MIN (:61)
constexpr double   kLn10Div10 = 0.230258509299404568401799145468436420760110148862877297603; // utils/maths.hpp:41
constexpr double   kLowest = -1.7976931348623157e308;

OCT_DEVICE uint32_t ld8(const uint8_t* p) { return *p; }
OCT_DEVICE uint64_t ld64u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
OCT_DEVICE bool is_acgt(uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
OCT_DEVICE uint32_t base_code(uint32_t c) { return (c >> 1) & 3u; }   // A 0, C 1, T 2, G 3

template <class T>
OCT_DEVICE uint32_t upper_bound_idx(const T* a, uint32_t n, T v)   // largest i in [0,n) with a[i] <= v (a[0] <= v assumed)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid; else hi = mid; }
    return lo;
}

// The same for a kernel whose consecutive threads hold consecutive values (one thread per pair / output): the wave searches ONCE, for its first
// value and in scalar registers; every lane then steps forward from there (a haplotype owns at least a read's worth of consecutive pairs, so
// mostly not at all). `first` must be wave-uniform and <= v. A many-region batch has tens of thousands of haplotypes: 16 dependent loads per
// thread otherwise.
template <class T>
OCT_DEVICE uint32_t upper_bound_near(const T* a, uint32_t n, T first, T v)
{
    uint32_t i = upper_bound_idx(a, n, first);
    while (i + 1 < n && a[i + 1] <= v) ++i;
    return i;
}
OCT_DEVICE uint64_t wave_first_index(uint64_t base)       // base + index of this wave's first thread in the grid, in scalar registers
{
    const uint32_t wave = hw::readfirstlane(hw::thread_idx() >> 6);
    return base + (uint64_t)hw::block_idx() * hw::block_dim() + (uint64_t)wave * 64u;
}

// ------------------------------------------------------------------------------------------------------------------
// per-read flags, per-haplotype-base DP tables
// ------------------------------------------------------------------------------------------------------------------
OCT_DEVICE void read_flags_wave(const DevBatch& b, uint32_t r, uint32_t lane)      // one wave per read: 64 bases per trip (a thread per read walked its 150 bases one
{                                                                                  // dependent load after the other: 25 us of a region-sized upload)
    if (r >= b.n_reads) return;
    const uint32_t ro = b.roff[r], T = b.roff[r + 1] - ro;
    uint32_t bad = 0;
    for (uint32_t i = lane; i < T; i += 64) bad |= is_acgt(b.rbases[ro + i]) ? 0u : 1u;
    const uint64_t any = hw::ballot(bad != 0);
    if (lane == 0) b.racgt[r] = any ? 0 : 1;
}

OCT_DEVICE uint32_t cap_of(uint32_t r, uint32_t h, uint32_t m, uint32_t p)
{
    // cost cap for read base r at a haplotype position: 0 on a match, the SNV prior if r equals the mask base, else none;
    // the read quality is min'ed in by the DP (update_match_state, simd_pair_hmm.hpp:121-132)
    if (r == h) return 0;
    return r == m ? p : 255u;
}

// Read-side operand of the fast-cost DP kernels, once per read instead of once per task (a read meets every haplotype of its region and
// ~1.4 candidate positions each): row r, entry j describes read position t = j - band as {v_perm selector of the base's cap byte | 0x0c00,
// quality << 16}; before the read and after its end {0x0d = "no cap", max_quality_score_ 64} (simd_pair_hmm.hpp:60,260,280). The DP kernel
// interleaves two rows into its LDS records with two v_perm per entry.
OCT_DEVICE void read_record_thread(const DevBatch& b, uint64_t g)
{
    const uint32_t r = (uint32_t)(g / b.rrec_stride), j = (uint32_t)(g % b.rrec_stride);
    if (r >= b.n_reads) return;
    const uint32_t ro = b.roff[r], T = b.roff[r + 1] - ro;
    const int32_t t = (int32_t)j - b.band;
    const bool in = t >= 0 && (uint32_t)t < T;
    const uint32_t base = in ? (uint32_t)b.rbases[ro + t] : 0u;
    const uint32_t sel = in ? base_code(base) : 0x0du, q = in ? (uint32_t)b.rquals[ro + t] : 64u;
    if (b.rrecW) {      // k_dp_mw, one task per band row: byte 0 the v_perm selector of the base's cap byte, bits 8-16 the raw base as k_dp_wide pads it, byte 3 the quality
        b.rrecW[g] = sel | (in ? base : (t < 0 ? 0x100u : (uint32_t)'0')) << 8 | q << 24;
        return;
    }
    b.rrec[g] = sel | 0x0c00u | q << 16;
}

// Once per batch (HaplotypeLikelihoodModel::reset analogue): the per-base DP tables; the workgroups past `table_blocks` set the per-read
// "pure ACGT" flags, those past `table_blocks + flag_blocks` build the read record rows, in the same launch.
OCT_DEVICE void hap_tables_block(const DevBatch& b, uint32_t n_bases, uint32_t table_blocks, uint32_t flag_blocks, uint32_t blk, uint32_t n_blk)   // workgroup `blk` of `n_blk`
{
    // the last workgroup also clears the step's counters (+ error key + overflow flag behind them), so that the first run after an upload needs no memset launch
    if (blk + 1 == n_blk) for (uint32_t i = hw::thread_idx(); i < kStatSlots * kStatStride + 2; i += hw::block_dim()) b.stats[i] = 0ull;
    if (blk >= table_blocks + flag_blocks) { read_record_thread(b, (uint64_t)(blk - table_blocks - flag_blocks) * hw::block_dim() + hw::thread_idx()); return; }
    if (blk >= table_blocks) { read_flags_wave(b, (blk - table_blocks) * (hw::block_dim() / 64) + (hw::thread_idx() >> 6), hw::thread_idx() & 63u); return; }
    const uint32_t g = blk * hw::block_dim() + hw::thread_idx();
    if (g >= n_bases) return;
    const uint32_t h = b.hbases[g], mf = b.maskF[g], mr = b.maskR[g];
    const uint32_t pf = (uint8_t)b.priorF[g], pr = (uint8_t)b.priorR[g];
    const uint32_t gg = ((uint32_t)(uint8_t)b.go[g] << 2) | ((uint32_t)(uint8_t)b.ge[g] << 18);   // {go<<2, ge<<2} as int16 halves
    const uint32_t A = 'A', C = 'C', T = 'T', G = 'G';                                             // column order = base_code
    b.tabFastF[g] = make_uint2(cap_of(A, h, mf, pf) | cap_of(C, h, mf, pf) << 8 | cap_of(T, h, mf, pf) << 16 | cap_of(G, h, mf, pf) << 24, gg);
    b.tabFastR[g] = make_uint2(cap_of(A, h, mr, pr) | cap_of(C, h, mr, pr) << 8 | cap_of(T, h, mr, pr) << 16 | cap_of(G, h, mr, pr) << 24, gg);
    const uint32_t isn = h == 'N' ? 1u : 0u;
    b.tabGenF[g] = make_uint2(h | mf << 8 | pf << 16 | isn << 24, gg);
    b.tabGenR[g] = make_uint2(h | mr << 8 | pr << 16 | isn << 24, gg);
    if (!is_acgt(h) || mf == '0' || mr == '0') {
        const uint32_t hap = upper_bound_idx(b.hoff, b.n_haps + 1, g);
        hw::atomic_and_u32(&b.hclean[hap], 0u);
    }
}
OCT_KERNEL(k_hap_tables)(DevBatch b, uint32_t n_bases, uint32_t table_blocks, uint32_t flag_blocks) { hap_tables_block(b, n_bases, table_blocks, flag_blocks, hw::block_idx(), hw::grid_dim()); }

// ------------------------------------------------------------------------------------------------------------------
// candidate mapping positions: 6-mer voting (utils/kmer_mapper.hpp)
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kKmer = 6, kKmerBins = 4096;                 // mapperKmerSize (haplotype_likelihood_array.hpp:103), num_kmers(6)

OCT_DEVICE uint32_t kmer_code(uint32_t c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; }   // perfect_hash :25-39 (everything else 0)
OCT_DEVICE uint32_t spread16(uint32_t x)                       // bit i of the low 16 -> bit 2i
{
    x = (x | x << 8) & 0x00ff00ffu; x = (x | x << 4) & 0x0f0f0f0fu; x = (x | x << 2) & 0x33333333u; return (x | x << 1) & 0x55555555u;
}
OCT_DEVICE uint32_t kmer_hash6(const uint8_t* s)               // perfect_kmer_hash<6> :43-53: sum of 4^j * code(base j)
{
    uint32_t h = 0;
    for (uint32_t j = 0; j < kKmer; ++j) h |= kmer_code(s[j]) << (2 * j);
    return h;
}

// make_kmer_hash_table (:85-106) for every haplotype: one workgroup per haplotype, CSR bins (order inside a bin does not
// affect the vote counts). LDS: 4096 counters + 256 scan slots.
constexpr uint32_t kHashSegment = 1024;                                       // bases per wave: a long read is hashed by several waves (DevBatch::hash_segs each; a 13 kb read was ~200 dependent trips of one wave)
OCT_DEVICE void read_hash_wave(const DevBatch& b, uint32_t w, uint32_t lane)   // compute_kmer_hashes<6> (:57-69): wave w = segment w % hash_segs of read w / hash_segs (no search for the read a base belongs to)
{
    const uint32_t r = w / b.hash_segs, seg = w % b.hash_segs;
    if (r >= b.n_reads) return;
    const uint32_t ro = b.roff[r], T = b.roff[r + 1] - ro;
    const uint32_t s0 = seg * kHashSegment, s1 = s0 + kHashSegment;             // this wave's positions [s0, s1)
    // k_kmer_map_lanes reads a read's hashes eight at a time, one lane per read: a row per read, 16-byte aligned, the entries behind the last k-mer = 4096
    // ("no k-mer": occupancy 0, equals no haplotype hash), so that its loop needs neither bounds tests nor unaligned loads. The other mappers read rhash.
    if (b.rhash) { for (uint32_t q = s0 + lane; q + kKmer <= T && q < s1; q += 64) b.rhash[ro + q] = (uint16_t)kmer_hash6(b.rbases + ro + q); }     // (the wave-per-pair mappers; the lane mapper reads the code rows below)
    if (b.rcode) {      // the read's 2-bit codes, 16 per dword (DevBatch::rcode): 64 bases per round, two ballots, lanes 0-3 store a dword each
        uint32_t* tile = b.rcode + ((size_t)(r >> 6) * b.rcode_words) * 64 + (r & 63u);
        for (uint32_t t0 = s0; t0 < b.rcode_words * 16 && t0 < s1; t0 += 64) {
            const uint32_t t = t0 + lane, c = t < T ? kmer_code(b.rbases[ro + t]) : 0u;
            const uint64_t b0 = hw::ballot((c & 1u) != 0), b1 = hw::ballot((c & 2u) != 0);
            const uint32_t j = (t0 >> 4) + lane;
            if (lane < 4 && j < b.rcode_words) tile[(size_t)j * 64] = spread16((uint32_t)(b0 >> (16 * lane)) & 0xffffu) | spread16((uint32_t)(b1 >> (16 * lane)) & 0xffffu) << 1;
        }
    }
}

// The workgroups past `n_hap_blocks` (one slice only) compute the read hashes of the whole batch in the same launch, four reads each.
OCT_DEVICE void kmer_tables_block(const DevBatch& b, uint32_t hap0, uint32_t n_hap_blocks, uint32_t blk, uint32_t* smem_words)
{
    if (blk >= n_hap_blocks) { read_hash_wave(b, (blk - n_hap_blocks) * (hw::block_dim() / 64) + (hw::thread_idx() >> 6), hw::thread_idx() & 63u); return; }
    uint32_t* hist = smem_words;                 // [4096]
    uint32_t* part = hist + kKmerBins;           // [256]
    const uint32_t h = hap0 + blk, tid = hw::thread_idx(), nt = hw::block_dim();
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho, nk = Lh >= kKmer ? Lh - kKmer + 1 : 0;
    for (uint32_t i = tid; i < kKmerBins; i += nt) hist[i] = 0;
    hw::block_sync();
    for (uint32_t p = tid; p < nk; p += nt) {
        const uint32_t hsh = kmer_hash6(b.hbases + ho + p);
        hw::atomic_add_lds_u32(&hist[hsh], 1u);
        b.hhash[ho + p] = (uint16_t)hsh;                              // the haplotype's own hash sequence (k_kmer_map's exact-count shortcut)
    }
    hw::block_sync();
    // exclusive scan of the 4096 counters: 16 per thread (256 threads = 4 waves); inside a wave by shuffles, across the waves through four
    // LDS words - two workgroup barriers instead of seventeen (a many-region batch runs this once per haplotype: 49 k workgroups per step)
    const uint32_t per = kKmerBins / 256, lane = tid & 63u, wv = tid >> 6;
    uint32_t cnt16[kKmerBins / 256], sum = 0;
    if (tid < 256) for (uint32_t i = 0; i < per; ++i) { cnt16[i] = hist[tid * per + i]; sum += cnt16[i]; }
    uint32_t incl = sum;
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = hw::shfl(incl, (int)(lane >= d ? lane - d : lane)); if (lane >= d) incl += o; }
    if (tid < 256 && lane == 63) part[wv] = incl;
    hw::block_sync();
    if (tid < 256) {
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wv; ++w) run += part[w];
        for (uint32_t i = 0; i < per; ++i) {
            const uint32_t c = cnt16[i]; hist[tid * per + i] = run | c << 16;                  // start and occupancy in one word (haplotypes are < 65,536 bases)
            run += c;
        }
    }
    hw::block_sync();
    if (b.bin32) {                                                                              // the table of k_kmer_map / k_kmer_map_lanes, 16 bytes per thread per store: whole lines
        uint4* dst = (uint4*)(b.bin32 + (size_t)h * kKmerBins); const uint4* src = (const uint4*)hist;
        for (uint32_t i = tid; i < kKmerBins / 4; i += nt) dst[i] = src[i];
    }
    hw::block_sync();
    for (uint32_t p = tid; p < nk; p += nt) {
        const uint32_t slot = hw::atomic_add_lds_u32(&hist[b.hhash[ho + p]], 1u) & 0xffffu;   // the start field counts up; it cannot reach the occupancy above it
        b.bin_idx[ho + slot] = (uint16_t)p;
    }
}
OCT_KERNEL(k_kmer_tables)(DevBatch b, uint32_t hap0, uint32_t n_hap_blocks) { OCT_DYN_SMEM(smem); kmer_tables_block(b, hap0, n_hap_blocks, hw::block_idx(), (uint32_t*)smem); }
// A one-shot call's first step: the batch's per-base DP tables, read flags and read records (k_hap_tables, otherwise part of the upload) in the SAME launch as the k-mer tables and
// read hashes - a region server's device batch is a chain of dependent launches, and these two do not depend on each other. Workgroups [0, n_kmer_blocks): k_kmer_tables' roles.
OCT_KERNEL(k_tables)(DevBatch b, uint32_t hap0, uint32_t n_hap_blocks, uint32_t n_kmer_blocks, uint32_t n_bases, uint32_t table_blocks, uint32_t flag_blocks)
{
    OCT_DYN_SMEM(smem);
    if (hw::block_idx() < n_kmer_blocks) kmer_tables_block(b, hap0, n_hap_blocks, hw::block_idx(), (uint32_t*)smem);
    else hap_tables_block(b, n_bases, table_blocks, flag_blocks, hw::block_idx() - n_kmer_blocks, hw::grid_dim() - n_kmer_blocks);
}

// compute_kmer_hashes<6> (:57-69) for every read, once per batch like the reference (haplotype_likelihood_array.cpp:118-131):
// rhash[roff[r] + q] for q <= T - 6.
// map_query_to_target (:120-159): one wave per (haplotype, read) pair; the workgroup keeps the haplotype's bins in LDS and its
// four waves stride over a chunk of the region's reads. Votes go to per-wave LDS counters; a 64-lane batch whose votes all fall on
// one diagonal (the normal case: the read's true offset) is merged into a single add.
constexpr uint32_t kMapPad = 520;          // sentinel entries behind the haplotype's hash sequence: q + d never needs a bounds test (q < 512 in k_kmer_map_lanes, q <= 255 in k_kmer_map; d < nk)
inline uint32_t kmer_map_lds_bytes(uint32_t lh_cap) { return (kKmerBins + 1) * 2 + 2 + (2 * ((lh_cap + 1) & ~1u) + kMapPad) * 2 + kKmerBins + 4 + kBlockWaves * (lh_cap + 64) * 4; }

// map_query_to_target's vote (:128-144) and its output (:145-157) for ONE (haplotype, read) pair by a whole wave: the path of the pairs the exact shortcuts of
// k_kmer_map / k_kmer_map_lanes cannot decide. `rh` = the read's hashes, `counts` = the wave's nk + 64 diagonal counters in LDS (zero on entry and on return).
template <class HashAt>      // hash_at(q) = the read's 6-mer hash at q (q < nq): from its hash row (k_kmer_map, k_kmer_map_big's callers) or cut out of its 2-bit code words (k_kmer_map_lanes)
OCT_DEVICE void kmer_count_votes_wave(const DevBatch& b, uint64_t e, HashAt hash_at, uint32_t nq, uint32_t nk, const uint16_t* bins, const uint16_t* idx,
                                      uint32_t* counts, uint32_t lane, uint32_t max_pos)
{
        uint32_t hq_next = lane < nq ? hash_at(lane) : 0;                   // software pipeline: next batch's hashes are in flight
        for (uint32_t q0 = 0; q0 < nq; q0 += 64) {
            const uint32_t q = q0 + lane;
            const bool valid = q < nq;
            const uint32_t hq = hq_next;
            if (q0 + 64 < nq) hq_next = q + 64 < nq ? hash_at(q + 64) : 0;
            const uint32_t b0 = bins[hq], n = valid ? (uint32_t)bins[hq + 1] - b0 : 0;
            for (uint32_t j = 0; hw::ballot(j < n) != 0; ++j) {
                const uint32_t ti = j < n ? idx[b0 + j] : 0;
                const bool vote = j < n && ti >= q;                    // :130
                const uint32_t d = ti - q;                             // mapping_begin :131
                const uint64_t voters = hw::ballot(vote);
                if (voters == 0) continue;
                const uint32_t src = (uint32_t)__builtin_ctzll(voters);
                const uint32_t d0 = hw::readlane(d, src);
                if (hw::ballot(vote && d != d0) == 0) {                // every vote on one diagonal: one add
                    if (lane == src) hw::atomic_add_lds_u32(&counts[d0], (uint32_t)__builtin_popcountll(voters));
                } else if (vote) {
                    hw::atomic_add_lds_u32(&counts[d], 1u);            // ++mapping_counts[mapping_begin], :132
                }
            }
        }
        hw::wave_lds_fence();
        // max_hit_count, then the ascending positions that reach it, at most max_pos of them (:145-157). The first 8 chunks of the
        // counters are read once into registers (and cleared: reset_mapping_counts :115-118) and reused by the output pass.
        uint32_t cr[8];
        uint32_t mx = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t d = (uint32_t)q * 64 + lane;
            cr[q] = d < nk ? counts[d] : 0;
            if (d < nk) counts[d] = 0;
            mx = cr[q] > mx ? cr[q] : mx;
        }
        for (uint32_t d = 512 + lane; d < nk; d += 64) { const uint32_t c = counts[d]; mx = c > mx ? c : mx; }
        mx = hw::wave_max_u32(mx);
        uint32_t n_out = 0;
        auto emit = [&](uint32_t d, uint32_t c) {
            const bool is = mx > 0 && d < nk && c == mx;
            const uint64_t mask = hw::ballot(is);
            const uint32_t rank = n_out + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (is && rank < max_pos) b.pos[e * (uint64_t)max_pos + rank] = d;
            n_out += (uint32_t)__builtin_popcountll(mask);
        };
#pragma unroll
        for (int q = 0; q < 8; ++q) { if ((uint32_t)q * 64 < nk) emit((uint32_t)q * 64 + lane, cr[q]); }
        for (uint32_t d0 = 512; d0 < nk; d0 += 64) {
            const uint32_t d = d0 + lane;
            const uint32_t c = d < nk ? counts[d] : 0;
            if (d < nk) counts[d] = 0;
            emit(d, c);
        }
        if (n_out > max_pos) n_out = max_pos;
        if (lane == 0) b.npos[e] = (uint8_t)n_out;
        hw::wave_lds_fence();
}

template <int ROUNDS>      // 64-lane rounds that hold the k-mers of the batch's longest read (3 for 150-base reads, at most 4 for the shortcut)
OCT_KERNEL(k_kmer_map)(DevBatch b, const uint32_t* blk_hap, const uint32_t* blk_read0, uint32_t lh_cap, uint32_t reads_per_block)
{
    OCT_DYN_SMEM(smem);
    uint16_t* bins = (uint16_t*)smem;                                  // [4097]
    uint16_t* idx = bins + kKmerBins + 2;                              // [lh_cap rounded to even]
    uint16_t* hh = idx + ((lh_cap + 1) & ~1u);                         // [lh_cap rounded to even] the haplotype's hash at every position
    uint8_t* occ = (uint8_t*)(hh + ((lh_cap + 1) & ~1u) + kMapPad);    // [4096 + 1] bin occupancy, capped at 255 (only 0 / 1 / more matter); entry 4096 = 0 is
                                                                       // the "hash" of lanes beyond a read's last k-mer
    uint32_t* counts_all = (uint32_t*)(occ + kKmerBins + 4);           // [waves][lh_cap + 64]
    const uint32_t tid = hw::thread_idx(), lane = tid & 63;
    const uint32_t wave = hw::readfirstlane(tid >> 6);                 // wave-uniform BY CONSTRUCTION, and the compiler must know it: read index, read offsets and
                                                                       // k-mer count then live in SGPRs (scalar loads, scalar branches) instead of per-lane copies
    const uint32_t h = blk_hap[hw::block_idx()], r_first = blk_read0[hw::block_idx()];
    const uint32_t g = b.hap_region[h];
    const uint32_t reg_r0 = b.reg_read0[g], reg_r1 = b.reg_read0[g + 1];
    const uint32_t r_end = r_first + reads_per_block < reg_r1 ? r_first + reads_per_block : reg_r1;
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho, nk = Lh >= kKmer ? Lh - kKmer + 1 : 0;   // table.second
    {   // bins and their occupancies out of the start | count << 16 table, four bins per 16-byte load (staging is per workgroup: it must stay small
        // beside the ~100 instructions a decided read costs)
        const uint4* src = (const uint4*)(b.bin32 + (size_t)h * kKmerBins);
        for (uint32_t i = tid; i < kKmerBins / 4; i += kBlockWaves * 64) {
            const uint4 v = src[i];
            const uint32_t c0 = v.x >> 16, c1 = v.y >> 16, c2 = v.z >> 16, c3 = v.w >> 16;
            *(uint2*)(bins + 4 * i) = make_uint2((v.x & 0xffffu) | v.y << 16, (v.z & 0xffffu) | v.w << 16);
            *(uint32_t*)(occ + 4 * i) = (c0 < 255 ? c0 : 255u) | (c1 < 255 ? c1 : 255u) << 8 | (c2 < 255 ? c2 : 255u) << 16 | (c3 < 255 ? c3 : 255u) << 24;
        }
        if (tid == 0) { bins[kKmerBins] = (uint16_t)nk; *(uint32_t*)(occ + kKmerBins) = 0; }
    }
    for (uint32_t i = tid; i < nk; i += kBlockWaves * 64) { idx[i] = b.bin_idx[ho + i]; hh[i] = b.hhash[ho + i]; }
    for (uint32_t i = nk + tid; i < ((lh_cap + 1) & ~1u) + kMapPad; i += kBlockWaves * 64) hh[i] = 0xffffu;   // no read hash equals it
    uint32_t* counts = counts_all + wave * (lh_cap + 64);
    for (uint32_t d = lane; d < nk + 64; d += 64) counts[d] = 0;
    hw::block_sync();
    const uint32_t max_pos = (uint32_t)b.max_pos;
    const uint64_t e_first = b.hap_pair_off[h];
    // The per-read work is a chain of dependent round trips (read offsets -> hashes -> bins -> bin entries); a wave walks its reads one
    // after the other, so the two global legs are software-pipelined across reads: while read i is processed, the hashes of read i + 1
    // and the offsets of read i + 2 are in flight.
    auto load_offsets = [&](uint32_t rr, uint32_t& ro_, uint32_t& nq_) {
        ro_ = 0; nq_ = 0;
        if (rr < r_end) { ro_ = b.roff[rr]; const uint32_t T_ = b.roff[rr + 1] - ro_; nq_ = T_ >= kKmer ? T_ - kKmer + 1 : 0; }   // compute_kmer_hashes :57-69
    };
    auto load_hashes = [&](uint32_t ro_, uint32_t nq_, uint32_t (&hv)[ROUNDS]) {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) { const uint32_t q = (uint32_t)k * 64 + lane; hv[k] = q < nq_ ? (uint32_t)b.rhash[ro_ + q] : kKmerBins; }   // 4096 = "no k-mer": occupancy 0, matches nothing
    };
    uint32_t ro = 0, nq = 0, ro_n = 0, nq_n = 0, hq4[ROUNDS], hq4_n[ROUNDS];
    load_offsets(r_first + wave, ro, nq);
    load_offsets(r_first + wave + kBlockWaves, ro_n, nq_n);
    load_hashes(ro, nq, hq4);
    for (uint32_t r = r_first + wave; r < r_end; r += kBlockWaves) {
        const uint64_t e = e_first + (r - reg_r0);
        load_hashes(ro_n, nq_n, hq4_n);                                        // read i + 1 (its offsets arrived during read i - 1)
        uint32_t ro_nn, nq_nn;
        load_offsets(r + 2 * kBlockWaves, ro_nn, nq_nn);                       // read i + 2
        bool decided = false;
        // Exact shortcut for the usual case, a read whose k-mers agree on ONE diagonal: the most frequent first-bin-entry diagonal d is a
        // candidate; its true count is a plain comparison of the two hash sequences along d (no counters); any other diagonal collects at
        // most one vote from each read k-mer that has a bin entry off d. If count(d) exceeds the number of such k-mers, d alone holds
        // max_hit_count and the answer is {d}. Anything else (near-even indel splits, repeats) takes the counting path below.
        if (nq > 0 && nq <= 64 * ROUNDS && nq <= 255 && !b.map_count_only) {     // (vote counts travel in bytes)
            // The CU's one scalar ALU is what this kernel runs out of (ballots, popcounts, scalar selects: ~2 scalar per vector instruction
            // in the first version), so everything below stays per lane and is reduced over the wave with DPP ladders on the vector side.
            uint32_t n4[ROUNDS];
            uint32_t key_lo = 0xffffffffu, key_hi = 0;                       // first / last read k-mer that occurs in the haplotype: q << 12 | hash
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) {
                n4[k] = (uint32_t)occ[hq4[k]];
                const uint32_t key = ((uint32_t)k * 64 + lane) << 13 | hq4[k];
                const uint32_t lo = n4[k] ? key : 0xffffffffu, hi = n4[k] ? key : 0u;
                key_lo = lo < key_lo ? lo : key_lo;
                key_hi = hi > key_hi ? hi : key_hi;
            }
            uint32_t w0 = 0xffffffffu, w1 = 0xffffffffu;                     // the winning diagonal(s)
            // Exact votes of up to two diagonals = positions where the two hash sequences agree along them; `others` = read k-mers that have a bin
            // entry on neither, i.e. an upper bound for the votes of ANY other diagonal. If the better of the two beats that bound, the answer is
            // that diagonal - or both, ascending, when they tie (a read split evenly by an indel).
            auto decide = [&](uint32_t dA, uint32_t dB) -> bool {           // dB may be "none" (0xffffffff)
                // per lane {votes on dA, votes on dB, k-mers with an entry on neither} in three bytes (a lane holds at most four k-mers)
                uint32_t acc = 0;
                const bool two = dB != 0xffffffffu && dB != dA;
#pragma unroll
                for (int k = 0; k < ROUNDS; ++k) {                           // no predicates: lanes beyond the read hold the non-hash 4096, positions
                    const uint32_t q = (uint32_t)k * 64 + lane;             // beyond the haplotype's last k-mer hold the sentinel 0xffff
                    const uint32_t ha = hh[q + dA];
                    const uint32_t onA = ha == hq4[k] ? 1u : 0u;
                    uint32_t onB = 0;
                    if (two) { const uint32_t hb = hh[q + dB]; onB = hb == hq4[k] ? 1u : 0u; }
                    acc += onA | onB << 8 | (n4[k] > onA + onB ? 1u << 16 : 0u);
                }
                const uint32_t tot = hw::wave_sum_u32(acc);
                const uint32_t cntA = tot & 0xffu, cntB = (tot >> 8) & 0xffu, others = tot >> 16;
                const uint32_t best = cntA > cntB ? cntA : cntB;
                if (best == 0 || best <= others) return false;
                if (cntA == cntB) { w0 = dA < dB ? dA : dB; w1 = dA < dB ? dB : dA; }       // only possible when `two`
                else { w0 = cntA > cntB ? dA : dB; w1 = 0xffffffffu; }
                return true;
            };
            // first attempt: the diagonals named by the first and by the last read k-mer that occur in the haplotype (first entry of their bins)
            uint32_t tried1 = 0xffffffffu, tried2 = 0xffffffffu;
            {
                const uint32_t kf = hw::wave_min_u32(key_lo), kl = hw::wave_max_u32(key_hi);
                if (kf != 0xffffffffu) {
                    const uint32_t qf = kf >> 13, ql = kl >> 13;
                    const uint32_t tf = idx[bins[kf & 0xfffu]], tl = idx[bins[kl & 0xfffu]];
                    if (tf >= qf) tried1 = tf - qf;
                    if (tl >= ql) tried2 = tl - ql;
                    if (tried1 == 0xffffffffu) { tried1 = tried2; tried2 = 0xffffffffu; }
                    if (tried1 != 0xffffffffu) decided = decide(tried1, tried2);
                }
            }
            // second attempt (a few percent of the pairs): the two most frequent first-entry diagonals over all read k-mers
            if (!decided) {
                uint32_t cand4[ROUNDS]; uint64_t rem4[ROUNDS]; uint32_t remaining = 0;
#pragma unroll
                for (int k = 0; k < ROUNDS; ++k) {
                    const uint32_t q = (uint32_t)k * 64 + lane;
                    const uint32_t t0 = idx[bins[hq4[k]] < nk ? bins[hq4[k]] : 0];
                    cand4[k] = (n4[k] && t0 >= q) ? t0 - q : 0xffffffffu;
                    rem4[k] = hw::ballot(cand4[k] != 0xffffffffu);
                    remaining += (uint32_t)__builtin_popcountll(rem4[k]);
                }
                uint32_t c1 = 0, d1 = 0xffffffffu, c2 = 0, d2 = 0xffffffffu;
                for (int it = 0; it < 8 && remaining > c1; ++it) {
                    uint32_t d = 0; bool got = false;
#pragma unroll
                    for (int k = 0; k < ROUNDS; ++k) if (!got && rem4[k]) { d = hw::readlane(cand4[k], (uint32_t)__builtin_ctzll(rem4[k])); got = true; }
                    uint32_t c = 0;
#pragma unroll
                    for (int k = 0; k < ROUNDS; ++k) { const uint64_t same = hw::ballot(cand4[k] == d) & rem4[k]; rem4[k] &= ~same; c += (uint32_t)__builtin_popcountll(same); }
                    remaining -= c;
                    if (c > c1) { c2 = c1; d2 = d1; c1 = c; d1 = d; } else if (c > c2) { c2 = c; d2 = d; }
                }
                if (c1 > 0 && remaining <= c1 && !(d1 == tried1 && d2 == tried2) && !(d1 == tried2 && d2 == tried1)) decided = decide(d1, d2);
            }
            if (decided && lane == 0) {
                uint32_t n_w = 0;
                if (max_pos >= 1) { b.pos[e * (uint64_t)max_pos] = w0; n_w = 1; }
                if (w1 != 0xffffffffu && max_pos >= 2) { b.pos[e * (uint64_t)max_pos + 1] = w1; n_w = 2; }
                b.npos[e] = (uint8_t)n_w;
            }
        }
        if (b.map_stats && lane == 0) hw::atomic_add_u64(b.stats + (size_t)(hw::block_idx() % kStatSlots) * kStatStride + (decided ? 6 : 7), 1ull);   // OCT_PHMM_MAP_STATS: pairs decided by the shortcut / counted
        if (!decided) { const uint16_t* rh = b.rhash + ro; kmer_count_votes_wave(b, e, [rh](uint32_t q) -> uint32_t { return rh[q]; }, nq, nk, bins, idx, counts, lane, max_pos); }
        ro = ro_n; nq = nq_n; ro_n = ro_nn; nq_n = nq_nn;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) hq4[k] = hq4_n[k];
    }
}

// The same mapping with ONE LANE per (haplotype, read) pair for big batches: a workgroup of 256 lanes takes 256 reads of one haplotype (tables in LDS as above).
// A lane runs the exact shortcut of k_kmer_map on its own: candidates = the diagonals named by the first k-mer of the read's first eight and by the last of its
// last eight that occur in the haplotype (first entry of their bins); one pass over the read's hashes counts, per lane and in registers, the votes of both
// (positions where the two hash sequences agree), the k-mers with a bin entry on neither (the bound for any other diagonal's votes) and - a 6-mer hash being two
// bits per base - the BASE mismatches along the first candidate, which is what try_naive_evaluate asks next (DevBatch::pair_mm, read by k_classify). No
// counters, no ballots, no reductions: ~40 wave-instructions per pair instead of the ~270 (120 vector + 150 scalar) of the wave-per-pair form, whose bound is the CU's one scalar ALU.
// The pairs a lane cannot decide (a few percent: both probes miss, repeats, near-even indel splits) are counted afterwards by the whole wave, one after the
// other (kmer_count_votes_wave). Same votes, same output: tests/check_populate.py::assert_device_positions.
constexpr uint32_t kLaneMapThreads = 256;
constexpr uint32_t kLaneMapMaxKmers = 496;       // lane form up to here (500-base chunks of long reads: 495 k-mers)
// dwords per read of DevBatch::rcode: the pass takes 16 k-mers (one dword of codes and the next one's first five bases) per step and has the step after in flight
OCT_HD uint32_t rcode_row_words(uint32_t t_cap) { const uint32_t nq = t_cap >= kKmer ? t_cap - kKmer + 1 : 0; return (nq + 15) / 16 + 3; }
// dwords of the haplotype's packed codes / its repeated-k-mer bits in LDS: a diagonal's words start anywhere below nk / 16 and run on for a read's steps + 2
OCT_HD uint32_t lane_map_hap_words(uint32_t lh_cap) { return lh_cap / 16 + (kLaneMapMaxKmers + 15) / 16 + 8; }
inline uint32_t kmer_map_lanes_lds_bytes(uint32_t lh_cap)
{
    return (kKmerBins + 2) * 2 + ((lh_cap + 1) & ~1u) * 2 + kKmerBins + 4 + 2 * lane_map_hap_words(lh_cap) * 4 + kBlockWaves * (lh_cap + 64) * 4;
}

// bits 2i, i < n (n clamped to 0 ... 16): the k-mers (or bases) of one 16-entry step that lie before a limit
OCT_DEVICE uint32_t low_pairs_mask(int32_t n) { n = n < 0 ? 0 : n > 16 ? 16 : n; return n ? 0x55555555u >> (32 - 2 * n) : 0u; }
OCT_DEVICE uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }   // v_alignbit_b32, sh < 32

OCT_KERNEL(k_kmer_map_lanes)(DevBatch b, const uint32_t* blk_hap, const uint32_t* blk_read0, uint32_t lh_cap)
{
    OCT_DYN_SMEM(smem);
    uint16_t* bins = (uint16_t*)smem;                                  // [4097]
    uint16_t* idx = bins + kKmerBins + 2;                              // [lh_cap rounded to even]
    uint8_t* occ = (uint8_t*)(idx + ((lh_cap + 1) & ~1u));             // [4096 + 1] bin occupancy capped at 255; entry 4096 = 0
    uint32_t* hc = (uint32_t*)(occ + kKmerBins + 4);                   // [lane_map_hap_words] the haplotype's 2-bit codes, 16 per dword, zero behind its last base
    uint32_t* mu = hc + lane_map_hap_words(lh_cap);                    // [lane_map_hap_words] bit 2(p & 15) of word p / 16: the k-mer at haplotype position p occurs more than once in the haplotype
    uint32_t* counts_all = mu + lane_map_hap_words(lh_cap);            // [waves][lh_cap + 64] (the counting path of undecided pairs)
    const uint32_t tid = hw::thread_idx(), lane = tid & 63;
    const uint32_t wave = hw::readfirstlane(tid >> 6);
    const uint32_t h = blk_hap[hw::block_idx()], r_first = blk_read0[hw::block_idx()];
    const uint32_t g = b.hap_region[h];
    const uint32_t reg_r0 = b.reg_read0[g], reg_r1 = b.reg_read0[g + 1];
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho, nk = Lh >= kKmer ? Lh - kKmer + 1 : 0;
    {
        const uint4* src = (const uint4*)(b.bin32 + (size_t)h * kKmerBins);
        for (uint32_t i = tid; i < kKmerBins / 4; i += kLaneMapThreads) {
            const uint4 v = src[i];
            const uint32_t c0 = v.x >> 16, c1 = v.y >> 16, c2 = v.z >> 16, c3 = v.w >> 16;
            *(uint2*)(bins + 4 * i) = make_uint2((v.x & 0xffffu) | v.y << 16, (v.z & 0xffffu) | v.w << 16);
            *(uint32_t*)(occ + 4 * i) = (c0 < 255 ? c0 : 255u) | (c1 < 255 ? c1 : 255u) << 8 | (c2 < 255 ? c2 : 255u) << 16 | (c3 < 255 ? c3 : 255u) << 24;
        }
        if (tid == 0) { bins[kKmerBins] = (uint16_t)nk; *(uint32_t*)(occ + kKmerBins) = 0; }
    }
    for (uint32_t i = tid; i < nk; i += kLaneMapThreads) idx[i] = b.bin_idx[ho + i];
    const uint32_t hap_words = lane_map_hap_words(lh_cap);
    // the haplotype's codes: 64 bases per wave and round, two ballots, lanes 0-3 store a dword each
    for (uint32_t p0 = wave * 64; p0 < hap_words * 16; p0 += kLaneMapThreads) {
        const uint32_t p = p0 + lane, c = p < Lh ? kmer_code(b.hbases[ho + p]) : 0u;
        const uint64_t b0 = hw::ballot((c & 1u) != 0), b1 = hw::ballot((c & 2u) != 0);
        if (lane < 4) hc[(p0 >> 4) + lane] = spread16((uint32_t)(b0 >> (16 * lane)) & 0xffffu) | spread16((uint32_t)(b1 >> (16 * lane)) & 0xffffu) << 1;
    }
    uint32_t* counts = counts_all + wave * (lh_cap + 64);
    for (uint32_t d = lane; d < nk + 64; d += 64) counts[d] = 0;
    hw::block_sync();
    for (uint32_t p0 = wave * 64; p0 < hap_words * 16; p0 += kLaneMapThreads) {       // (after the barrier: occ is complete)
        const uint32_t p = p0 + lane;
        const uint64_t m = hw::ballot(p < nk && occ[b.hhash[ho + (p < nk ? p : 0)]] >= 2);
        if (lane < 4) mu[(p0 >> 4) + lane] = spread16((uint32_t)(m >> (16 * lane)) & 0xffffu);
    }
    hw::block_sync();
#if defined(OCT_MAP_PROBE) && OCT_MAP_PROBE == 3
    { const uint32_t rp_ = r_first + tid; if (rp_ < reg_r1) b.npos[b.hap_pair_off[h] + (rp_ - reg_r0)] = 0; }
    return;                                                            // TIMING PROBE (tools/sessions_r06): the workgroup's staging alone (no candidates: every pair keeps its original position only)
#endif
    const uint32_t max_pos = (uint32_t)b.max_pos, none = 0xffffffffu;
    const uint32_t r = r_first + tid;
    const bool live = r < reg_r1;
    const uint64_t e = b.hap_pair_off[h] + (uint64_t)((live ? r : reg_r0) - reg_r0);
    uint32_t nq = 0, T = 0;
    if (live) { const uint32_t ro = b.roff[r]; T = b.roff[r + 1] - ro; nq = T >= kKmer ? T - kKmer + 1 : 0; }
    const bool eligible = live && nq > 0 && nq <= kLaneMapMaxKmers && !b.map_count_only;
    const uint32_t rr = live ? r : reg_r0;
    const uint32_t* rc = b.rcode + ((size_t)(rr >> 6) * b.rcode_words) * 64 + (rr & 63u);     // this lane's column of its 64-read tile: dword j at rc[j * 64]
    // Three probes - the read's first, its middle and its last k-mer that occur exactly ONCE in the haplotype (a k-mer of a homopolymer or repeat names the
    // diagonal of its first copy, mostly the wrong one), each naming the diagonal of its bin's one entry - give the two candidates. A diagonal can only beat the bound below with more than half of the read's k-mers on it, and such a stretch of the read holds the middle
    // k-mer: dA = the middle probe's diagonal (a read whose two ends lie across two indels from each other is decided by it), dB = the first probe's where that
    // differs, else the last one's (a read split by one indel: both of its diagonals, as k_kmer_map's first attempt takes them). Eight k-mers per step (the 26 bits
    // behind base 8c of the read's codes: two coalesced dword loads and a funnel shift), until every lane of the wave has found its own: mostly one step per probe (a read's tail is where its errors sit).
    uint32_t dA = none, dB = none, dC = none;
    {
        const uint32_t nq_e = eligible ? nq : 0u;
        auto chunk_bits = [&](uint32_t c) -> uint32_t { return funnel(rc[(size_t)((c >> 1) + 1) * 64], rc[(size_t)(c >> 1) * 64], (c & 1u) * 16u); };   // bases 8c ... 8c + 15
        auto scan_up = [&](uint32_t c0, uint32_t& q_found, uint32_t& h_found) {                   // the first k-mer from chunk c0 on that occurs once in the haplotype
            for (uint32_t c = c0; hw::ballot(q_found == none && c * 8 < nq_e) != 0; ++c) {
                const uint32_t v = chunk_bits(c);
                uint32_t qj = none, hj = 0;
#pragma unroll
                for (int j = 7; j >= 0; --j) { const uint32_t q = c * 8 + (uint32_t)j, hq = (v >> (2 * j)) & 0xfffu; if (q < nq_e && occ[hq] == 1) { qj = q; hj = hq; } }
                if (q_found == none && qj != none) { q_found = qj; h_found = hj; }
            }
        };
        uint32_t qf = none, hqf = 0, qm = none, hqm = 0, ql = none, hql = 0;
        scan_up(0, qf, hqf);
        scan_up((nq_e >> 1) >> 3, qm, hqm);
        const uint32_t c_last = hw::wave_max_u32(nq_e ? (nq_e - 1) >> 3 : 0u);
        for (uint32_t c = c_last + 1; c-- > 0 && hw::ballot(ql == none && c * 8 < nq_e) != 0; ) {
            const uint32_t v = chunk_bits(c);
            uint32_t qj = none, hj = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint32_t q = c * 8 + (uint32_t)j, hq = (v >> (2 * j)) & 0xfffu; if (q < nq_e && occ[hq] == 1) { qj = q; hj = hq; } }
            if (ql == none && qj != none) { ql = qj; hql = hj; }
        }
        uint32_t dF = none, dM = none, dL = none;
        if (qf != none) { const uint32_t t = idx[bins[hqf]]; if (t >= qf) dF = t - qf; }
        if (qm != none) { const uint32_t t = idx[bins[hqm]]; if (t >= qm) dM = t - qm; }
        if (ql != none) { const uint32_t t = idx[bins[hql]]; if (t >= ql) dL = t - ql; }
        dA = dM != none ? dM : dF != none ? dF : dL;
        dB = (dF != none && dF != dA) ? dF : dL;
        if (dB == dA) dB = none;
        // a third candidate where all three probes disagree - a read across two indels. (With a gather per k-mer and candidate - the first form of this kernel - the third stream
        // cost more than the pairs it decided saved, 2.29 -> 2.51 ms per launch; bit-parallel it is 14 instructions and two gathers per 16 k-mers, and the counting path it
        // spares is two thirds of the kernel's time: profiles/r06_s06_mapper_phases.txt.)
        dC = (dL != none && dL != dA && dL != dB && dB != none) ? dL : none;
        if (!eligible) { dA = none; dB = none; dC = none; }
    }
    // ONE pass over the read, bit-parallel (round 6): 16 k-mers per step and diagonal. With read and haplotype as 2-bit codes, 16 per dword, the bases of a step against a
    // diagonal are one funnel shift of two haplotype words and one XOR; a k-mer's six bases agree where none of six neighbouring base bits is set (five funnel shifts and ORs
    // over this step's and the next one's bits); the votes of a diagonal are a population count. What any OTHER diagonal can collect is bounded from above without a look-up per
    // k-mer: a read k-mer that sits on neither candidate may vote elsewhere (counted, whether or not the haplotype holds it at all), one that sits on a candidate votes elsewhere
    // only if the haplotype holds it more than once - a bit per haplotype position (mu). The bound is never below the exact one the first form counted with a gather per k-mer
    // (occ[hash] > onA + onB), so "best beats it" still proves the answer; the few pairs it no longer decides are counted by the wave below like every undecided pair. Per pair:
    // 4 LDS gathers and ~55 vector instructions per 16 k-mers (first form: 3 gathers and ~20 instructions per k-mer, 2.8 bank-conflict cycles per gather).
    const uint32_t limA = (eligible && dA != none && nk > dA) ? (nq < nk - dA ? nq : nk - dA) : 0u;     // k-mers q < lim have a haplotype k-mer beside them on the diagonal
    const uint32_t limB = (eligible && dB != none && nk > dB) ? (nq < nk - dB ? nq : nk - dB) : 0u;
    const uint32_t limC = (eligible && dC != none && nk > dC) ? (nq < nk - dC ? nq : nk - dC) : 0u;
    const uint32_t wA = limA ? dA >> 4 : 0u, shA = limA ? 2u * (dA & 15u) : 0u, wB = limB ? dB >> 4 : 0u, shB = limB ? 2u * (dB & 15u) : 0u;
    const uint32_t wC = limC ? dC >> 4 : 0u, shC = limC ? 2u * (dC & 15u) : 0u;
    uint32_t cntA = 0, cntB = 0, cntC = 0, others = 0, mm = 0, mm_pos = 0;
#if defined(OCT_MAP_PROBE) && OCT_MAP_PROBE == 2
    const uint32_t nq_wave = 0;                                        // TIMING PROBE: staging + probes, no pass
#else
    const uint32_t nq_wave = hw::wave_max_u32(limA ? nq : 0u);
#endif
    {
        auto mism = [](uint32_t x) -> uint32_t { return (x | x >> 1) & 0x55555555u; };                 // bit 2i: base i differs
        auto kmer_bad = [](uint32_t m0, uint32_t m1) -> uint32_t {                                        // bit 2i: some base of the k-mer at i differs (m1: the next 16 bases)
            const uint32_t a_lo = m0 | funnel(m1, m0, 2), a_hi = m1 | m1 >> 2;
            return a_lo | funnel(a_hi, a_lo, 4) | funnel(a_hi, a_lo, 8);
        };
        uint32_t r0 = rc[0], r1 = rc[64];
        uint32_t hAl = hc[wA + 1], hBl = hc[wB + 1], uAl = mu[wA], uBl = mu[wB], hCl = hc[wC + 1], uCl = mu[wC];
        uint32_t mA0 = mism(r0 ^ funnel(hAl, hc[wA], shA)), mB0 = mism(r0 ^ funnel(hBl, hc[wB], shB)), mC0 = mism(r0 ^ funnel(hCl, hc[wC], shC));
        const bool any_c = hw::ballot(limC != 0) != 0;                                                 // (most waves have no lane with a third candidate: they skip its stream)
        for (uint32_t j = 0; j * 16 < nq_wave; ++j) {
            const uint32_t r2 = rc[(size_t)(j + 2) * 64];                                               // (the row's slack covers it)
            const uint32_t hAh = hc[wA + j + 2], hBh = hc[wB + j + 2], uAh = mu[wA + j + 1], uBh = mu[wB + j + 1];
            const uint32_t mA1 = mism(r1 ^ funnel(hAh, hAl, shA)), mB1 = mism(r1 ^ funnel(hBh, hBl, shB));
            const int32_t q0 = (int32_t)(j * 16);
            const uint32_t okA = ~kmer_bad(mA0, mA1) & low_pairs_mask((int32_t)limA - q0), okB = ~kmer_bad(mB0, mB1) & low_pairs_mask((int32_t)limB - q0);
            cntA += (uint32_t)__builtin_popcount(okA); cntB += (uint32_t)__builtin_popcount(okB);
            uint32_t rep = (okA & funnel(uAh, uAl, shA)) | (okB & funnel(uBh, uBl, shB));            // on a candidate, and the haplotype holds the k-mer elsewhere too
            uint32_t okC = 0;
            if (any_c) {
                const uint32_t hCh = hc[wC + j + 2], uCh = mu[wC + j + 1];
                const uint32_t mC1 = mism(r1 ^ funnel(hCh, hCl, shC));
                okC = ~kmer_bad(mC0, mC1) & low_pairs_mask((int32_t)limC - q0);
                cntC += (uint32_t)__builtin_popcount(okC);
                rep |= okC & funnel(uCh, uCl, shC);
                mC0 = mC1; hCl = hCh; uCl = uCh;
            }
            others += (uint32_t)__builtin_popcount(rep | (low_pairs_mask((int32_t)nq - q0) & ~(okA | okB | okC)));   // ... or on no candidate: may vote anywhere, if the haplotype holds the k-mer at all
            const uint32_t t = mA0 & low_pairs_mask((int32_t)T - q0);                                  // base mismatches along the first candidate (DevBatch::pair_mm)
            mm += (uint32_t)__builtin_popcount(t);
            mm_pos = t ? (uint32_t)q0 + ((uint32_t)__builtin_ctz(t) >> 1) : mm_pos;
            mA0 = mA1; mB0 = mB1; hAl = hAh; hBl = hBh; uAl = uAh; uBl = uBh; r1 = r2;
        }
        {   // the bases behind the last step's sixteen (a read's last five bases belong to no k-mer of their own)
            const int32_t q0 = (int32_t)(((nq_wave + 15) / 16) * 16);
            const uint32_t t = mA0 & low_pairs_mask((int32_t)T - q0);
            mm += (uint32_t)__builtin_popcount(t);
            mm_pos = t ? (uint32_t)q0 + ((uint32_t)__builtin_ctz(t) >> 1) : mm_pos;
        }
    }
    bool decided = false;
    uint32_t w0 = none, w1 = none, mm_word = 0;
    const uint32_t bestAB = cntA > cntB ? cntA : cntB, best = bestAB > cntC ? bestAB : cntC;
    if (eligible && dA != none) {
        if (best != 0 && best > others) {
            decided = true;
            // the candidates that reach the maximum, ascending (map_query_to_target's output order, :145-157); equal counts need distinct diagonals, which the candidates are
            uint32_t w2 = none;
            const uint32_t a = cntA == best ? dA : none, bb = cntB == best ? dB : none, c = cntC == best ? dC : none;
            const uint32_t lo = a < bb ? a : bb, hi = a < bb ? bb : a;                 // ("none" = 0xffffffff sorts last)
            w0 = lo < c ? lo : c; const uint32_t rest = lo < c ? c : lo;
            w1 = hi < rest ? hi : rest; w2 = hi < rest ? rest : hi;
            if (w1 == none && w0 == dA) mm_word = mm == 0 ? 1u << 14 : mm == 1 ? (2u << 14 | mm_pos) : 3u << 14;
            uint32_t n_w = 0;
            if (max_pos >= 1) { b.pos[e * (uint64_t)max_pos] = w0; n_w = 1; }
            if (w1 != none && max_pos >= 2) { b.pos[e * (uint64_t)max_pos + 1] = w1; n_w = 2; }
            if (w2 != none && max_pos >= 3) { b.pos[e * (uint64_t)max_pos + 2] = w2; n_w = 3; }
            b.npos[e] = (uint8_t)n_w;
        }
    }
#if defined(OCTPHMM_SIM) && defined(OCT_DEBUG_MAP)
    if (live && !decided) fprintf(stderr, "undecided: nq %u dA %d dB %d cntA %u cntB %u others %u nk %u\n", nq, (int)dA, (int)dB, cntA, cntB, others, nk);
#endif
    if (live && b.pair_mm) b.pair_mm[e] = (uint16_t)mm_word;
    // the undecided pairs of this wave, one after the other, by the whole wave
    uint64_t todo = hw::ballot(live && !decided);
#if defined(OCT_MAP_PROBE)
    if (live && !decided) b.npos[e] = 0;
    todo = 0;                                                          // TIMING PROBE: no counting path (undecided pairs keep their original position only)
#endif
    const uint64_t all = hw::ballot(live);
    if (b.map_stats && lane == 0) {
        unsigned long long* st = b.stats + (size_t)(hw::block_idx() % kStatSlots) * kStatStride;
        hw::atomic_add_u64(st + 6, (unsigned long long)__builtin_popcountll(all & ~todo)); hw::atomic_add_u64(st + 7, (unsigned long long)__builtin_popcountll(todo));
    }
    while (todo) {
        const uint32_t src = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t r_u = hw::readlane(r, src), nq_u = hw::readlane(nq, src);
        const uint64_t e_u = b.hap_pair_off[h] + (uint64_t)(r_u - reg_r0);
        // (the read's hashes cut out of its code words: lane q's twelve bits lie in dwords q / 16 and q / 16 + 1 of the read's column - a round of 64 k-mers touches five dwords)
        const uint32_t* rc_u = b.rcode + ((size_t)(r_u >> 6) * b.rcode_words) * 64 + (r_u & 63u);
        kmer_count_votes_wave(b, e_u, [rc_u](uint32_t q) -> uint32_t { const uint32_t j = q >> 4; return funnel(rc_u[(size_t)(j + 1) * 64], rc_u[(size_t)j * 64], 2u * (q & 15u)) & 0xfffu; },
                              nq_u, nk, bins, idx, counts, lane, max_pos);
    }
}

// Long haplotypes (bins + per-wave counters no longer fit LDS beside each other): one workgroup per (haplotype, read) pair, the diagonal counters alone in LDS, bins read from
// global memory. Same votes, same result as k_kmer_map; for the long-read configuration. Round 6: the counters are 16-bit halves of LDS words (a diagonal collects at most one
// vote per read k-mer, and reads are shorter than 32,768 bases): twice the workgroups per CU (ccs2048x12: 5.4 -> 4.9 ms per launch), and haplotypes up to the 16-bit bin tables' own
// limit of 65,535 bases (40 k before). Measured and not kept (profiles/r06_s17_*): the bin starts and entries staged in LDS beside the counters (4.9 ms as well: 40 KB of staging per
// pair), and staging + merging a wave's votes where its 64 k-mers agree on a diagonal as kmer_count_votes_wave does (6.2 ms: every 6-mer of a 16 kb haplotype has ~4 bin entries,
// the lanes' j-th entries never agree).
OCT_HD uint32_t kmer_map_big_lds_bytes(uint32_t lh_cap) { return ((lh_cap + 2) / 2) * 4 + 64; }
#ifndef OCT_BIG_MAP_U
#define OCT_BIG_MAP_U 4
#endif
constexpr uint32_t kBigMapU = OCT_BIG_MAP_U;      // k-mers in flight per thread
OCT_KERNEL(k_kmer_map_big)(DevBatch b, uint64_t pair0)
{
    OCT_DYN_SMEM(smem);
    uint32_t* counts = (uint32_t*)smem;                                // [(lh_cap + 2) / 2] two 16-bit counters per word: diagonal d in half d & 1 of word d / 2
    __shared__ uint32_t s_max, s_nout;
    const uint64_t e = pair0 + hw::block_idx();
    const uint32_t tid = hw::thread_idx(), nt = hw::block_dim();
    const uint32_t h = upper_bound_idx(b.hap_pair_off, b.n_haps + 1, e);
    const uint32_t g = b.hap_region[h];
    const uint32_t r = b.reg_read0[g] + (uint32_t)(e - b.hap_pair_off[h]);
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho, nk = Lh >= kKmer ? Lh - kKmer + 1 : 0;
    const uint32_t ro = b.roff[r], T = b.roff[r + 1] - ro, nq = T >= kKmer ? T - kKmer + 1 : 0;
    const uint32_t* bins = b.bin32 + (size_t)h * kKmerBins;             // start | occupancy << 16 per bin (k_kmer_tables)
    const uint16_t* idx = b.bin_idx + ho;
    for (uint32_t w = tid; w < (nk + 1) / 2 + 1; w += nt) counts[w] = 0;
    if (tid == 0) { s_max = 0; s_nout = 0; }
    hw::block_sync();
    // A k-mer's votes are a chain of three dependent gathers (its hash -> its bin -> the bin's entries): one word per bin (start | occupancy << 16), a bin's first four entries in ONE
    // 8-byte load (they are neighbours in bin_idx; a 16 kb haplotype's bins hold ~4), kBigMapU k-mers per thread in flight, stage by stage. What the kernel runs out of, though, is the
    // LDS: ~48,000 counter atomics per 12 kb read on random banks (SQ_LDS_IDX_ACTIVE = 0.93 x SQ_BUSY_CYCLES, 38 % of it bank conflicts: profiles/r06_s24_long_read_pmc.txt) - the
    // loads' forms above moved the launch by 3 %, counting the diagonals a wave agrees on with ballots instead of atomics made it 17 % slower (profiles/EXPERIMENTS.md).
    auto vote_d = [&](uint32_t d) { hw::atomic_add_lds_u32(&counts[d >> 1], 1u << (16 * (d & 1u))); };                                                               // :130-132
    for (uint32_t q0 = tid; q0 < nq; q0 += nt * kBigMapU) {
        uint32_t hq[kBigMapU], j0[kBigMapU], nj[kBigMapU]; uint64_t ent[kBigMapU];
#pragma unroll
        for (uint32_t u = 0; u < kBigMapU; ++u) { const uint32_t q = q0 + u * nt; hq[u] = q < nq ? (uint32_t)b.rhash[ro + q] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < kBigMapU; ++u) { const uint32_t w = bins[hq[u]]; j0[u] = w & 0xffffu; nj[u] = q0 + u * nt < nq ? w >> 16 : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < kBigMapU; ++u) ent[u] = ld64u((const uint8_t*)(idx + j0[u]));       // (bin_idx has eight spare entries behind the last haplotype's)
#pragma unroll
        for (uint32_t u = 0; u < kBigMapU; ++u) {
            const uint32_t q = q0 + u * nt;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) { const uint32_t ti = (uint32_t)(ent[u] >> (16 * k)) & 0xffffu; if (k < nj[u] && ti >= q) vote_d(ti - q); }
            for (uint32_t j = 4; j < nj[u]; ++j) { const uint32_t ti = idx[j0[u] + j]; if (ti >= q) vote_d(ti - q); }
        }
    }
    hw::block_sync();
    auto cnt_of = [&](uint32_t d) -> uint32_t { return (counts[d >> 1] >> (16 * (d & 1u))) & 0xffffu; };
    uint32_t mx = 0;
    for (uint32_t w = tid; w < (nk + 1) / 2; w += nt) { const uint32_t c = counts[w], lo = c & 0xffffu, hi = c >> 16; mx = lo > mx ? lo : mx; mx = hi > mx ? hi : mx; }
    if (mx) hw::atomic_max_lds_u32(&s_max, mx);
    hw::block_sync();
    mx = s_max;
    // the ascending offsets that reach the maximum, at most max_pos of them (:145-157): every thread owns a contiguous stretch of
    // diagonals, the stretches' match counts are prefix-summed, and a thread writes its matches while their rank is below max_pos
    const uint32_t lane = tid & 63u;
    __shared__ uint32_t s_part[16];                                    // per wave: its threads' matches (the workgroup has at most 16 waves)
    // (an ODD number of counter words per thread: with 32 - a 16 kb haplotype over 256 threads - the 64 lanes of a wave read 64 words of ONE bank, pass after pass, and this tail
    // was most of the kernel: 3.2 of ccs2048x12's 16.5 ms)
    const uint32_t per = 2 * ((((nk + 1) / 2 + nt - 1) / nt) | 1u), d0 = tid * per < nk ? tid * per : nk, d1 = d0 + per < nk ? d0 + per : nk;
    uint32_t mine = 0;
    if (mx > 0) for (uint32_t d = d0; d < d1; ++d) mine += cnt_of(d) == mx ? 1u : 0u;
    uint32_t inc = mine;                                               // inclusive prefix inside the wave by shuffles, across the waves through s_part (one thread walked all 256 counts before)
    for (uint32_t dd = 1; dd < 64; dd <<= 1) { const uint32_t o = hw::shfl(inc, (int)(lane >= dd ? lane - dd : lane)); if (lane >= dd) inc += o; }
    if (lane == 63) s_part[tid >> 6] = inc;
    hw::block_sync();
    uint32_t rank = inc - mine, all = 0;
    for (uint32_t w = 0; w < nt / 64; ++w) { const uint32_t c = s_part[w]; if (w < (tid >> 6)) rank += c; all += c; }
    if (tid == 0) s_nout = all;
    hw::block_sync();
    if (mine) for (uint32_t d = d0; d < d1 && rank < (uint32_t)b.max_pos; ++d) if (cnt_of(d) == mx) b.pos[e * (uint64_t)b.max_pos + rank++] = d;
    if (tid == 0) b.npos[e] = (uint8_t)(s_nout < (uint32_t)b.max_pos ? s_nout : (uint32_t)b.max_pos);
}

// ------------------------------------------------------------------------------------------------------------------
// candidate enumeration + scalar fast path
// ------------------------------------------------------------------------------------------------------------------
OCT_DEVICE uint32_t first_mismatch(const uint8_t* a, const uint8_t* b, uint32_t from, uint32_t n)
{
    uint32_t i = from;
    // 16 bytes per step, four loads in flight: a thread's pass over a read is a chain of round trips to memory, one per step (32 bytes per step: two spilled registers in k_classify)
    while (i + 16 <= n) {
        const uint64_t x0 = ld64u(a + i) ^ ld64u(b + i), x1 = ld64u(a + i + 8) ^ ld64u(b + i + 8);
        if (x0 | x1) return x0 ? i + (uint32_t)(__builtin_ctzll(x0) >> 3) : i + 8 + (uint32_t)(__builtin_ctzll(x1) >> 3);
        i += 16;
    }
    while (i + 8 <= n) {
        const uint64_t x = ld64u(a + i) ^ ld64u(b + i);
        if (x) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
        i += 8;
    }
    while (i < n) { if (a[i] != b[i]) return i; ++i; }
    return n;
}
OCT_DEVICE bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t n) { return first_mismatch(a, b, 0, n) == n; }

// hmm::detail::try_naive_evaluate, pair_hmm.hpp:278-319. Returns true when handled; *pen = phred penalty.
// try_naive_one: its second half, for a read with exactly ONE mismatch against the haplotype at `pos`, at read position i1 (:290-318).
OCT_DEVICE bool try_naive_one(const uint8_t* truth, uint32_t Lh, const uint8_t* target, uint32_t T, const uint8_t* quals, uint32_t pos, uint32_t i1,
                              const int8_t* go, const int8_t* ge, const uint8_t* mask, const int8_t* prior,
                              uint32_t lhs, uint32_t rhs, int32_t* pen)
{
    const uint8_t* tr = truth + pos;
    const uint64_t idx = (uint64_t)i1 + pos;
    if (idx < (uint64_t)lhs || idx >= ((uint64_t)Lh - (uint64_t)rhs)) { *pen = 0; return true; }   // is_in_flank :206-214
    uint32_t mp = quals[i1];
    if (mask[idx] == target[i1]) { const uint32_t p = (uint8_t)prior[idx]; if (p < mp) mp = p; }     // get_mismatch_penalty :250-263
    const int32_t gop = go[idx];
    if ((int32_t)mp <= gop) { *pen = (int32_t)mp; return true; }
    if (bytes_equal(target + i1 + 1, tr + i1, T - i1 - 1)) { *pen = gop; return true; }            // :305
    if (bytes_equal(target + i1, tr + i1 + 1, T - i1)) { *pen = gop; return true; }                // :309
    if ((int32_t)mp <= gop + (int32_t)ge[idx]) { *pen = (int32_t)mp; return true; }                // :313
    return false;
}
OCT_DEVICE bool try_naive(const uint8_t* truth, uint32_t Lh, const uint8_t* target, uint32_t T, const uint8_t* quals, uint32_t pos,
                          const int8_t* go, const int8_t* ge, const uint8_t* mask, const int8_t* prior,
                          uint32_t lhs, uint32_t rhs, int32_t* pen)
{
    const uint8_t* tr = truth + pos;
    const uint32_t i1 = first_mismatch(target, tr, 0, T);
    if (i1 == T) { *pen = 0; return true; }
    if (first_mismatch(target, tr, i1 + 1, T) != T) return false;
    return try_naive_one(truth, Lh, target, T, quals, pos, i1, go, ge, mask, prior, lhs, rhs, pen);
}

OCT_DEVICE bool pos_in_range(uint64_t p, uint32_t T, uint32_t Lh, uint32_t B)   // num_out_of_range_bases(...) == 0, model.cpp:187-207
{
    return p >= B && p + T + B <= Lh;
}
OCT_DEVICE int32_t num_out_of_range(uint64_t p, uint32_t T, uint32_t Lh, uint32_t B)
{
    if (p < B) return (int32_t)(B - p);
    const uint64_t e = p + T + B;
    if (e > Lh) return (int32_t)Lh - (int32_t)e;
    return 0;
}

OCT_DEVICE uint64_t wave_sum(uint64_t v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    for (int m = 1; m < 64; m <<= 1) {
        const uint64_t o = (uint64_t)hw::shfl_xor(lo, m) | ((uint64_t)hw::shfl_xor(hi, m) << 32);
        const uint64_t s = (((uint64_t)hi << 32) | lo) + o;
        lo = (uint32_t)s; hi = (uint32_t)(s >> 32);
    }
    return ((uint64_t)hi << 32) | lo;
}

OCT_DEVICE uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// Exact de-duplication of pairs (section below k_classify). What decides a pair's result beyond the read: fast-path minimum, task classes,
// haplotype length, and per DP task its position and the canonical window it starts in. k_classify leaves a 32-bit hash of these per pair
// (0 = no DP task); `pair_views_same` compares two pairs of one read field by field.
OCT_DEVICE uint64_t pair_hash_task(uint64_t acc, uint32_t p, uint32_t canon) { return mix64(acc ^ ((uint64_t)p << 32 | canon)); }
OCT_DEVICE uint32_t pair_hash_final(uint64_t acc, uint32_t cls, int32_t best, uint32_t Lh, uint32_t mask)
{
    const uint32_t k = (uint32_t)mix64(acc ^ mix64((uint64_t)cls << 32 | (uint32_t)best) ^ Lh) & mask;
    return k ? k : 1u;
}
struct PairView { uint32_t cls; int32_t best; uint32_t Lh; uint64_t e; uint32_t ho; };
OCT_DEVICE uint32_t pair_slot_pos(const DevBatch& b, const PairView& v, uint32_t slot) { return slot < (uint32_t)b.max_pos ? b.pos[v.e * (uint64_t)b.max_pos + slot] : b.pair_extra[v.e]; }
OCT_DEVICE bool pair_views_same(const DevBatch& b, const PairView& v, const PairView& w)
{
    if (v.cls != w.cls || v.best != w.best || v.Lh != w.Lh) return false;
    for (uint32_t slot = 0; slot <= (uint32_t)b.max_pos; ++slot) {
        if (!((v.cls >> (2 * slot)) & 3u)) continue;
        const uint32_t p = pair_slot_pos(b, v, slot), q = pair_slot_pos(b, w, slot);
        if (p != q) return false;
        const uint32_t off = p > (uint32_t)b.band ? p - (uint32_t)b.band : 0;
        if (b.canon[v.ho + off] != b.canon[w.ho + off]) return false;
    }
    return true;
}

// Workgroup-local exclusive scan of one uint4 per thread over the 256 threads of k_classify / k_dedup_verify (the first half of the task-count scan, see
// k_scan_finish): returns the thread's exclusive prefix inside its workgroup, *total = the workgroup's sum. sh: 4 words of LDS.
OCT_DEVICE uint4 block256_scan_excl(uint4 v, uint4* sh, uint4* total)
{
    const uint32_t tid = hw::thread_idx(), lane = tid & 63u, wv = tid >> 6;
    uint4 inc = v;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const int src = (int)(lane >= d ? lane - d : lane);
        const uint4 o = make_uint4(hw::shfl(inc.x, src), hw::shfl(inc.y, src), hw::shfl(inc.z, src), hw::shfl(inc.w, src));
        if (lane >= d) inc = make_uint4(inc.x + o.x, inc.y + o.y, inc.z + o.z, inc.w + o.w);
    }
    hw::block_sync();                                           // (an earlier scan's reads of sh are over)
    if (lane == 63) sh[wv] = inc;
    hw::block_sync();
    uint4 before = make_uint4(0, 0, 0, 0), all = make_uint4(0, 0, 0, 0);
    for (uint32_t w = 0; w < 4; ++w) {
        const uint4 t = sh[w];
        if (w < wv) before = make_uint4(before.x + t.x, before.y + t.y, before.z + t.z, before.w + t.w);
        all = make_uint4(all.x + t.x, all.y + t.y, all.z + t.z, all.w + t.w);
    }
    *total = all;
    return make_uint4(before.x + inc.x - v.x, before.y + inc.y - v.y, before.z + inc.z - v.z, before.w + inc.w - v.w);
}
// store the counts of pairs [pair0, pair1) of this workgroup scanned tile-locally (+ the extra entry at pair1, whose thread counts nothing), and the tile's total
OCT_DEVICE void store_scanned_local(uint4 mine, uint64_t e, uint64_t pair0, uint64_t pair1, uint4* cnt, uint4* tile_sums, uint4* sh)
{
    uint4 total;
    const uint4 ex = block256_scan_excl(mine, sh, &total);
    if (e <= pair1) cnt[e - pair0] = ex;
    if (hw::thread_idx() == 0) tile_sums[hw::block_idx()] = total;
}

// Pass 1: one thread per (read, haplotype) pair. Runs the candidate-position logic and the scalar fast path, leaves the
// best fast-path penalty in pair_best, classifies every remaining candidate as score-only or traceback DP.
// A batch is processed in slices of whole haplotypes (pairs [pair0, pair1)); `cnt` is the slice's own scan array (pair1 - pair0 + 1 entries).
// tile_sums != null: the counts are stored scanned across the workgroup (store_scanned_local; the grid then covers pair1 itself, the scan's extra entry)
#if defined(OCT_CLASSIFY_WAVES) && !defined(OCTPHMM_SIM)
__attribute__((amdgpu_waves_per_eu(OCT_CLASSIFY_WAVES, OCT_CLASSIFY_WAVES)))      // (A/B builds: tools/build_variant.sh)
#endif
OCT_KERNEL(k_classify)(DevBatch b, uint64_t pair0, uint64_t pair1, uint4* cnt, uint4* cnt_late, uint4* tile_sums, uint4* tile_sums_late)
{
    __shared__ uint4 sh_scan[4];
    uint4 my_cnt = make_uint4(0, 0, 0, 0), my_late = make_uint4(0, 0, 0, 0);
    const uint64_t e = pair0 + (uint64_t)hw::block_idx() * hw::block_dim() + hw::thread_idx();
    const uint64_t e_wave = wave_first_index(pair0);
    unsigned long long st_cand = 0, st_fast = 0, st_score = 0, st_trace = 0, st_cells = 0, st_pairs = 0;
    if (e < pair1) {
        const uint32_t h = upper_bound_near(b.hap_pair_off, b.n_haps + 1, e_wave, e);
        const uint32_t g = b.hap_region[h];
        const uint32_t r = b.reg_read0[g] + (uint32_t)(e - b.hap_pair_off[h]);
        const uint32_t ro = b.roff[r], T = b.roff[r + 1] - ro, ho = b.hoff[h], Lh = b.hoff[h + 1] - ho;
        const uint32_t B = (uint32_t)b.band, lhs = b.reg_lhs[g], rhs = b.reg_rhs[g];
        const uint8_t* target = b.rbases + ro; const uint8_t* quals = b.rquals + ro; const uint8_t* truth = b.hbases + ho;
        const bool fwd = !b.rrev[r];
        const uint8_t* mask = (fwd ? b.maskF : b.maskR) + ho; const int8_t* prior = (fwd ? b.priorF : b.priorR) + ho;
        const int8_t* go = b.go + ho; const int8_t* ge = b.ge + ho;
        const uint64_t orig = (uint64_t)(b.rbegin[r] - b.hbegin[h]);                               // begin_distance, model.cpp:220
        const uint32_t* P = b.pos + e * (uint64_t)b.max_pos; const uint32_t npos = b.npos[e];
        const bool clean = b.racgt[r] && b.hclean[h];                                               // pure ACGT on both sides: equal 6-mer hashes <=> equal bases
        // the mapper's account of the base mismatches along the pair's one mapped position (DevBatch::pair_mm): slot 0 then needs no pass over the bases
        const uint32_t mm = b.pair_mm && clean && npos == 1 ? (uint32_t)b.pair_mm[e] : 0u, mm_state = mm >> 14, mm_i1 = mm & 0x3fffu;
        int32_t best = kNoScore; uint32_t cls = 0, n_score = 0, n_trace = 0, n_late = 0, extra = 0;
        bool orig_mapped = false, any = false;
        unsigned long long key = ~0ull;
        uint64_t dedup_acc = 0;                                                                     // k_dedup_match: hash over the DP tasks, in slot order
        auto visit = [&](uint32_t slot, uint32_t p) {
            ++st_cand;
            const uint32_t known = slot == 0 ? mm_state : 0u;
#if defined(OCTPHMM_SIM)
            if (known) {    // CPU suite only: the mapper's account must be what the bytes say at THIS position (a stale pair_mm beside other positions would turn a mismatch into penalty 0 silently - ADVICE r04)
                const uint32_t i1 = first_mismatch(target, truth + p, 0, T), i2 = i1 == T ? T : first_mismatch(target, truth + p, i1 + 1, T);
                const uint32_t want = i1 == T ? 1u : i2 == T ? 2u : 3u;
                if (want != known || (known == 2 && i1 != mm_i1) || !pos_in_range(p, T, Lh, 0)) { fprintf(stderr, "k_classify: pair_mm drifted from the bases (pair %llu, state %u, bytes say %u)\n", (unsigned long long)e, known, want); abort(); }
            }
#endif
            if (b.align_mode) {                                                                     // hmm::align, pair_hmm.hpp:861-872
                bool same = known ? known == 1 : true;                                              // try_naive_align :321-341
                if (!known) for (uint32_t tt = 0; tt < T; ++tt) if (target[tt] != truth[p + tt]) { same = false; break; }
                const unsigned long long order = slot == (uint32_t)b.max_pos ? 0ull : (unsigned long long)slot + 1;
                if (same) { ++st_fast; const unsigned long long k = order << 8 | 1ull; if (k < key) key = k; return; }
                st_cells += 2ull * B * (T + B);                                                     // simd_align always tracebacks (:806-810)
                cls |= 2u << (2 * slot); ++n_trace; ++st_trace;
                return;
            }
            int32_t pen = 0;
            const bool handled = known == 1 ? true : known == 3 ? false
                               : known == 2 ? try_naive_one(truth, Lh, target, T, quals, p, mm_i1, go, ge, mask, prior, lhs, rhs, &pen)
                               : try_naive(truth, Lh, target, T, quals, p, go, ge, mask, prior, lhs, rhs, &pen);
            if (handled) { ++st_fast; if (pen < best) best = pen; return; }
            const uint32_t off = p > B ? p - B : 0;                                                 // pair_hmm.hpp:735
            if ((uint64_t)off + T + 2 * B - 1 > Lh) return;                                         // :736-738 -> lowest()
            const bool adjusted = (uint64_t)p < (uint64_t)lhs + B || (uint64_t)p + T + B > (uint64_t)Lh - (uint64_t)rhs;   // :123-137
            st_cells += 2ull * B * (T + B);
            if (b.canon) dedup_acc = pair_hash_task(dedup_acc, p, b.canon[ho + off]);
            // class 3 = traceback needed for the right flank only (the window starts at or after the left flank's end): late traceback start
            if (adjusted && cnt_late && (uint64_t)lhs + B <= (uint64_t)p) { cls |= 3u << (2 * slot); ++n_late; ++st_trace; }
            else if (adjusted) { cls |= 2u << (2 * slot); ++n_trace; ++st_trace; } else { cls |= 1u << (2 * slot); ++n_score; ++st_score; }
        };
        for (uint32_t j = 0; j < npos; ++j) {                                                       // model.cpp:223-232
            const uint32_t p = P[j];
            if ((uint64_t)p == orig) orig_mapped = true;
            if (pos_in_range(p, T, Lh, B)) { any = true; visit(j, p); }
        }
        if (!orig_mapped && pos_in_range(orig, T, Lh, B)) { any = true; extra = (uint32_t)orig; visit((uint32_t)b.max_pos, extra); }   // :233-237
        if (!any) {                                                                                  // :238-256
            const int32_t min_shift = num_out_of_range(orig, T, Lh, B);
            uint64_t fin = orig; bool err = false;
            if (min_shift > 0) { fin += (uint64_t)min_shift; if (!pos_in_range(fin, T, Lh, B)) err = true; }
            else { const uint32_t left = (uint32_t)(-min_shift); if (orig >= left) fin -= left; else err = true; }
            if (err) hw::atomic_max_u64(b.err_key, ~(((unsigned long long)h << 32) | r));   // stored inverted: the first failing pair is the maximum, 0 = none
            else { extra = (uint32_t)fin; visit((uint32_t)b.max_pos, extra); }
        }
        b.pair_best[e] = best; b.pair_cls[e] = cls; b.pair_extra[e] = extra;
        if (b.canon) { b.pair_hash[e] = cls ? pair_hash_final(dedup_acc, cls, best, Lh, b.dedup_hash_mask) : 0u; b.pair_fast[e] = best; }
        if (b.align_mode) b.pair_key[e] = key;
        const bool generic = b.wide || !clean;
        my_cnt = generic ? make_uint4(0, 0, n_score, n_trace) : make_uint4(n_score, n_trace, 0, 0);
        my_late = generic ? make_uint4(0, n_late, 0, 0) : make_uint4(n_late, 0, 0, 0);
        if (!tile_sums) {
            cnt[e - pair0] = my_cnt;
            if (e + 1 == pair1) cnt[pair1 - pair0] = make_uint4(0, 0, 0, 0);                           // the scan's extra entry (totals land here)
            if (cnt_late) { cnt_late[e - pair0] = my_late; if (e + 1 == pair1) cnt_late[pair1 - pair0] = make_uint4(0, 0, 0, 0); }
        }
        st_pairs = 1;
    }
    if (tile_sums) {
        store_scanned_local(my_cnt, e, pair0, pair1, cnt, tile_sums, sh_scan);
        if (cnt_late) store_scanned_local(my_late, e, pair0, pair1, cnt_late, tile_sums_late, sh_scan);
    }
    st_cand = wave_sum(st_cand); st_fast = wave_sum(st_fast); st_score = wave_sum(st_score);
    st_trace = wave_sum(st_trace); st_cells = wave_sum(st_cells); st_pairs = wave_sum(st_pairs);
    if ((hw::thread_idx() & 63) == 0) {
        // counters are spread over kStatSlots cache lines (summed on the host) so that ~10^5 waves do not serialise on one line
        unsigned long long* st = b.stats + (size_t)(hw::block_idx() % kStatSlots) * kStatStride;
        if (st_cand)  hw::atomic_add_u64(st + 0, st_cand);
        if (st_fast)  hw::atomic_add_u64(st + 1, st_fast);
        if (st_score) hw::atomic_add_u64(st + 2, st_score);
        if (st_trace) hw::atomic_add_u64(st + 3, st_trace);
        if (st_cells) hw::atomic_add_u64(st + 4, st_cells);
        if (st_pairs) hw::atomic_add_u64(st + 5, st_pairs);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Exact de-duplication of (read, haplotype) pairs. The haplotypes of a region are a few edits apart, so many of them show a read the very
// same band window: same bases, same six penalty vectors, same place in a haplotype of the same length. Such pairs have the same
// candidates, the same fast-path penalties, the same DP tasks and the same result - the reference computes each of them again.
//   upload: canon[hoff[h] + off] = index of the first window of the region with the bytes of the window of haplotype h at off (length
//           t_cap + 2 B - 1, cut at the haplotype's end; found through a hash table, CONFIRMED byte by byte)
//   step:   k_classify leaves a hash per pair; k_dedup_match lets a read look at its pairs haplotype by haplotype and remember candidates;
//           k_dedup_verify compares: a pair whose fast-path minimum, task classes, positions, haplotype length and canonical windows all
//           equal those of an earlier pair of the same read drops its DP tasks and points at that pair (pair_rep); k_epilogue reads the
//           result there.
// Everything is decided by comparing the values themselves; hashes only find candidates.
// ------------------------------------------------------------------------------------------------------------------

// prefix[hoff[h] + h + x] = sum over y < x of mix(position y of haplotype h) * pw[y] (mod 2^64): one wave per haplotype, 64 positions per
// round, an inclusive scan across the lanes (shuffles on the two halves of the 64-bit sums) and a carry from round to round
OCT_DEVICE uint64_t shfl_u64(uint64_t v, uint32_t src) { return (uint64_t)hw::shfl((uint32_t)v, (int)src) | (uint64_t)hw::shfl((uint32_t)(v >> 32), (int)src) << 32; }
OCT_KERNEL(k_window_prefix)(DevBatch b, const uint64_t* pw, uint64_t* prefix)
{
    const uint32_t lane = hw::thread_idx() & 63u;
    const uint32_t h = hw::block_idx() * (hw::block_dim() / 64) + hw::readfirstlane(hw::thread_idx() >> 6);
    if (h >= b.n_haps) return;                                  // (whole waves)
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho;
    uint64_t* out = prefix + (size_t)ho + h;
    if (lane == 0) out[0] = 0;
    uint64_t carry = 0;
    for (uint32_t x0 = 0; x0 < Lh; x0 += 64) {
        const uint32_t x = x0 + lane;
        uint64_t sum = 0;
        if (x < Lh) {
            const uint64_t v = (uint64_t)b.hbases[ho + x] | (uint64_t)(uint8_t)b.go[ho + x] << 8 | (uint64_t)(uint8_t)b.ge[ho + x] << 16 | (uint64_t)b.maskF[ho + x] << 24
                             | (uint64_t)(uint8_t)b.priorF[ho + x] << 32 | (uint64_t)b.maskR[ho + x] << 40 | (uint64_t)(uint8_t)b.priorR[ho + x] << 48;
            sum = mix64(v + 1) * pw[x];
        }
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint64_t o = shfl_u64(sum, lane >= d ? lane - d : lane); if (lane >= d) sum += o; }
        if (x < Lh) out[x + 1] = carry + sum;
        carry += shfl_u64(sum, 63);
    }
}

// key of every window (never 0) and, per key, the smallest window index: open addressing, keys claimed with compare-and-swap
// (round 4: the table is one small table PER REGION - tab_base[g], tab_mask[g]: a power of two of slots around 1.25-2.5 x the region's windows - because a region's windows only
// ever meet each other: the threads of a wave work on one region's few hundred KB of slots, which stay in L2, instead of scattering atomics over one 800 MB table)
OCT_KERNEL(k_window_insert)(DevBatch b, const uint64_t* pwinv, const uint64_t* prefix, uint32_t n_bases, unsigned long long* wkey,
                            unsigned long long* tkeys, uint32_t* tvals, const uint32_t* tab_base, const uint32_t* tab_mask, const uint32_t* blk_hap, int phase)
{
    const uint32_t x = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (x >= n_bases) return;
    // blk_hap[workgroup] = the haplotype of the workgroup's first window (host-made): a binary search per wave over 49 k haplotype offsets was 16 dependent scalar loads, and under
    // the load of 300 k waves those were most of the kernel
    uint32_t h = blk_hap[hw::block_idx()];
    while (h + 1 < b.n_haps && b.hoff[h + 1] <= x) ++h;
    // Two launches: phase 0 = the windows of every region's FIRST haplotype, phase 1 = all the others. A region's haplotypes are on the chip at the same time, so in one
    // launch its ~20 copies of a window all found the slot empty and all swapped (30 M atomics for 19.6 M windows: the L2's atomic unit was the bound); after phase 0 most
    // windows of phase 1 find their key with a plain look and touch nothing.
    if (((h == 0 || b.hap_region[h - 1] != b.hap_region[h]) ? 0 : 1) != phase) return;
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho, off = x - ho;
    const uint32_t end = off + b.window_len < Lh ? off + b.window_len : Lh;
    const uint64_t* pre = prefix + (size_t)ho + h;
    const uint64_t sum = (pre[end] - pre[off]) * pwinv[off];                 // the window's polynomial, independent of where it starts
    const unsigned long long key = (mix64(sum ^ mix64((uint64_t)(end - off) << 32 | b.hap_region[h])) & ((uint64_t)b.dedup_hash_mask << 32 | b.dedup_hash_mask)) | 1ull;
    wkey[x] = key;
    const uint32_t g = b.hap_region[h], base = tab_base[g], tmask = tab_mask[g];
    // A key is met by as many windows as haplotypes share it (5-20 in a region of a few edits apart): one compare-and-swap and one atomic minimum per window were 39 M atomics on
    // ~3 M addresses, and the L2's atomic unit was the kernel's bound. A plain look first: a slot that already shows the key needs no swap, a value already below x no minimum (values
    // only fall; a stale look costs an atomic that changes nothing).
    for (uint32_t slot = (uint32_t)(key >> 20) & tmask; ; slot = (slot + 1) & tmask) {
        unsigned long long seen = hw::load_device_u64(tkeys + base + slot);
        if (seen == 0ull) seen = hw::atomic_cas_u64(tkeys + base + slot, 0ull, key);
        if (seen == 0ull || seen == key) { if (hw::load_device_u32(tvals + base + slot) > x) hw::atomic_min_u32(tvals + base + slot, x); break; }
    }
}

OCT_DEVICE bool window_bytes_equal(const DevBatch& b, uint32_t a, uint32_t c, uint32_t n)   // n bytes of all seven arrays from a and from c
{
    return bytes_equal(b.hbases + a, b.hbases + c, n) && bytes_equal((const uint8_t*)b.go + a, (const uint8_t*)b.go + c, n)
        && bytes_equal((const uint8_t*)b.ge + a, (const uint8_t*)b.ge + c, n) && bytes_equal(b.maskF + a, b.maskF + c, n)
        && bytes_equal((const uint8_t*)b.priorF + a, (const uint8_t*)b.priorF + c, n) && bytes_equal(b.maskR + a, b.maskR + c, n)
        && bytes_equal((const uint8_t*)b.priorR + a, (const uint8_t*)b.priorR + c, n);
}

// canon[x] = the table's window for x's key if its bytes (and region, and length) are those of x, else x itself. Two steps:
//   k_window_candidate  one thread per window: canon[x] = the table's window `first` when region and length agree (the CANDIDATE), else x
//   k_window_confirm    one wave per haplotype, its windows in order: consecutive windows share all but one position and, between two haplotypes that are a few
//                       edits apart, consecutive windows name consecutive candidates - so a window whose candidate continues its predecessor's (first == the
//                       predecessor's first + 1) is confirmed by ITS LAST POSITION alone once the predecessor is; only the head of such a run compares whole
//                       windows (7 arrays x window_len bytes - what every window did before: 25 GB of compares per upload of the 2,000-region stream,
//                       2.5 ms; now the runs' heads and 14 bytes per window).
// Still every byte of a shared window has been compared with its canonical twin's: hashes only propose.
OCT_DEVICE uint32_t window_len_at(const DevBatch& b, uint32_t off, uint32_t Lh) { return (off + b.window_len < Lh ? off + b.window_len : Lh) - off; }
OCT_KERNEL(k_window_candidate)(DevBatch b, uint32_t n_bases, const unsigned long long* wkey, const unsigned long long* tkeys, const uint32_t* tvals,
                               const uint32_t* tab_base, const uint32_t* tab_mask, const uint32_t* blk_hap)
{
    const uint32_t x = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (x >= n_bases) return;
    const unsigned long long key = wkey[x];
    uint32_t h = blk_hap[hw::block_idx()];
    while (h + 1 < b.n_haps && b.hoff[h + 1] <= x) ++h;
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho;
    const uint32_t g = b.hap_region[h], base = tab_base[g], tmask = tab_mask[g];
    uint32_t slot = (uint32_t)(key >> 20) & tmask;
    while (tkeys[base + slot] != key) slot = (slot + 1) & tmask;
    const uint32_t first = tvals[base + slot];
    uint32_t cand = x;
    if (first != x) {
        const uint32_t h2 = upper_bound_idx(b.hoff, b.n_haps + 1, first), ho2 = b.hoff[h2], Lh2 = b.hoff[h2 + 1] - ho2;
        if (window_len_at(b, x - ho, Lh) == window_len_at(b, first - ho2, Lh2) && b.hap_region[h] == b.hap_region[h2]) cand = first;
    }
    b.canon[x] = cand;
}
// Keys, table and candidates of a region with the table in LDS (round 4, step 7; replaces k_window_insert x 2 + k_window_candidate and their tables in global memory).
// A region's windows only ever meet each other and there are 10^3 - 10^5 of them: the 30 M device-scope atomics of the global tables (the XCDs' L2s are not coherent with each
// other, so every one of them travels to the memory side: ~20 G atomics/s, 1.5 ms per 2,000-region upload) become LDS atomics. Slot = tag << 32 | smallest window index (the tag:
// the key's high half; equal tags are taken for equal keys - k_window_confirm compares the bytes anyway, a false candidate only costs that window its sharing), ~0 = empty, so one
// 64-bit minimum both claims and lowers. A region with more windows than half the table is cut into residue classes of its keys (equal keys, same class), ONE WORKGROUP PER CLASS
// (`blocks`, host-made: region and class of every workgroup): each computes every key of the region and keeps its own - the first form ran a region's classes one after the other in
// one workgroup, and the few regions of 10^5 windows (13 classes) held the launch for milliseconds.
constexpr uint32_t kWinSlots = 16384, kWinPassWindows = 8192, kWinThreads = 1024, kWinHapStage = 1024;
// The key's class, 0 .. n_pass - 1: a multiply-high of the hashed tag. NOT `(tag * c >> 8) % n_pass`: where the compiler can see that both operands fit 24 bits it divides through a
// float reciprocal, and this toolchain's expansion (ROCm 7.2, gfx950) returns 0xffffff instead of n - 1 for 2.8 % of 24-bit dividends at n = 11 (and at 22, 23; quotients off by one
// likewise - tools/urem_probe.hip, profiles/r04_step8_urem24_probe.txt). Keys with that "class" belonged to no workgroup, their windows kept whatever canon[] held, and
// k_window_confirm followed the garbage: a memory fault on region-server batches holding a region of eleven classes. The simulator's host compiler divides exactly.
OCT_DEVICE uint32_t window_pass_of(unsigned long long key, uint32_t n_pass) { return (uint32_t)(((unsigned long long)((uint32_t)(key >> 32) * 0x9e3779b1u) * n_pass) >> 32); }
OCT_DEVICE uint32_t window_slot_of(unsigned long long key) { return ((uint32_t)key * 0x85ebca6bu >> 12) & (kWinSlots - 1); }
inline uint32_t window_passes(uint64_t n_windows) { return (uint32_t)((n_windows + kWinPassWindows - 1) / kWinPassWindows); }
inline size_t window_region_lds_bytes() { return (size_t)kWinSlots * 8 + (size_t)(kWinHapStage + 1) * 4; }
OCT_MAX_THREADS(1024) OCT_KERNEL(k_window_region)(DevBatch b, const uint64_t* pwinv, const uint64_t* prefix, const uint32_t* reg_hap0, const uint2* blocks)
{
    OCT_DYN_SMEM(smem);
    unsigned long long* tab = (unsigned long long*)smem;          // [kWinSlots]
    uint32_t* hoff_l = (uint32_t*)(tab + kWinSlots);              // [kWinHapStage + 1]: the region's haplotype offsets (regions with more haplotypes read them from memory)
    const uint32_t tid = hw::thread_idx(), nt = hw::block_dim();
    const uint32_t g = blocks[hw::block_idx()].x, pass = blocks[hw::block_idx()].y;
    const uint32_t h0 = reg_hap0[g], h1 = reg_hap0[g + 1];
    if (h0 == h1) return;
    const bool staged = h1 - h0 <= kWinHapStage;
    if (staged) for (uint32_t i = tid; i <= h1 - h0; i += nt) hoff_l[i] = b.hoff[h0 + i];
    for (uint32_t i = tid; i < kWinSlots; i += nt) tab[i] = ~0ull;
    hw::block_sync();
    auto hoff_at = [&](uint32_t hp) -> uint32_t { return staged ? hoff_l[hp - h0] : b.hoff[hp]; };
    const uint32_t lo = hoff_at(h0), hi = hoff_at(h1);
    const uint32_t n_pass = (hi - lo + kWinPassWindows - 1) / kWinPassWindows;
    // Every window's key (never 0), as k_window_insert made it, FOUR windows of a thread at a time: the haplotypes first (LDS), then the twelve loads of the four keys in one
    // basic block (addresses clamped instead of guarded, so that they are all in flight together - one window at a time, a thread waited three dependent round trips per
    // window and a workgroup of a 10^5-window region took a millisecond), then the table. The second loop makes the keys again instead of reading them back (two multiplies and two mixes per window against 8 B stored and loaded).
    constexpr uint32_t U = 4;
    struct Four { unsigned long long key[U]; uint32_t hap[U]; };
    auto keys_of = [&](uint32_t x0, uint32_t& h) -> Four {
        Four f; uint32_t ho[U], off[U], end[U];
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t x = x0 + u * nt < hi ? x0 + u * nt : hi - 1;
            while (hoff_at(h + 1) <= x) ++h;
            f.hap[u] = h; ho[u] = hoff_at(h); off[u] = x - ho[u];
            const uint32_t Lh = hoff_at(h + 1) - ho[u];
            end[u] = off[u] + b.window_len < Lh ? off[u] + b.window_len : Lh;
        }
        uint64_t pe[U], po[U], pi[U];
        for (uint32_t u = 0; u < U; ++u) { const uint64_t* pre = prefix + (size_t)ho[u] + f.hap[u]; pe[u] = pre[end[u]]; po[u] = pre[off[u]]; pi[u] = pwinv[off[u]]; }
        for (uint32_t u = 0; u < U; ++u) {
            const uint64_t sum = (pe[u] - po[u]) * pi[u];                            // the window's polynomial, independent of where it starts
            f.key[u] = (mix64(sum ^ mix64((uint64_t)(end[u] - off[u]) << 32 | g)) & ((uint64_t)b.dedup_hash_mask << 32 | b.dedup_hash_mask)) | 1ull;
        }
        return f;
    };
    {
        uint32_t h = h0;
        for (uint32_t x0 = lo + tid; x0 < hi; x0 += U * nt) {
            const Four f = keys_of(x0, h);
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t x = x0 + u * nt;
                const unsigned long long key = f.key[u];
                if (x >= hi || (n_pass > 1 && window_pass_of(key, n_pass) != pass)) continue;
                const unsigned long long word = (key & 0xffffffff00000000ull) | x;
                uint32_t slot = window_slot_of(key);
                for (uint32_t step = 0; step < kWinSlots; ++step, slot = (slot + 1) & (kWinSlots - 1)) {    // (a full table - more distinct keys in a class than slots - leaves the window to itself)
                    unsigned long long seen = tab[slot];
                    if (seen == ~0ull) seen = hw::atomic_cas_lds_u64(&tab[slot], ~0ull, word);
                    if (seen == ~0ull) break;                                                                // claimed
                    if ((seen >> 32) == (key >> 32)) { if (seen > word) hw::atomic_min_lds_u64(&tab[slot], word); break; }
                }
            }
        }
    }
    hw::block_sync();
    uint32_t h = h0;
    for (uint32_t x0 = lo + tid; x0 < hi; x0 += U * nt) {
        const Four f = keys_of(x0, h);                                               // (which of the four are ours, and their haplotypes)
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t x = x0 + u * nt;
            const unsigned long long key = f.key[u];
            if (x >= hi || (n_pass > 1 && window_pass_of(key, n_pass) != pass)) continue;
            uint32_t first = x, slot = window_slot_of(key);
            for (uint32_t step = 0; step < kWinSlots; ++step, slot = (slot + 1) & (kWinSlots - 1)) {
                const unsigned long long seen = tab[slot];
                if (seen == ~0ull) break;
                if ((seen >> 32) == (key >> 32)) { first = (uint32_t)seen; break; }
            }
            uint32_t cand = x;
            if (first != x) {                                                        // the candidate: same length (the region is the same by construction), k_window_confirm compares the bytes
                uint32_t a = h0, c = h1;                                             // the haplotype of `first`: the last one of the region that starts at or before it
                while (c - a > 1) { const uint32_t mid = (a + c) >> 1; if (hoff_at(mid) <= first) a = mid; else c = mid; }
                const uint32_t hx = f.hap[u], ho = hoff_at(hx), Lh = hoff_at(hx + 1) - ho, ho2 = hoff_at(a), Lh2 = hoff_at(a + 1) - ho2;
                if (window_len_at(b, x - ho, Lh) == window_len_at(b, first - ho2, Lh2)) cand = first;
            }
            b.canon[x] = cand;
        }
    }
}
OCT_DEVICE bool window_position_equal(const DevBatch& b, uint32_t a, uint32_t c)       // the seven arrays at a and at c
{
    return b.hbases[a] == b.hbases[c] && b.go[a] == b.go[c] && b.ge[a] == b.ge[c] && b.maskF[a] == b.maskF[c] && b.priorF[a] == b.priorF[c]
        && b.maskR[a] == b.maskR[c] && b.priorR[a] == b.priorR[c];
}
OCT_KERNEL(k_window_confirm)(DevBatch b)
{
    const uint32_t lane = hw::thread_idx() & 63u;
    const uint32_t h = hw::block_idx() * (hw::block_dim() / 64) + hw::readfirstlane(hw::thread_idx() >> 6);
    if (h >= b.n_haps) return;                                  // (whole waves)
    const uint32_t ho = b.hoff[h], Lh = b.hoff[h + 1] - ho;
    uint32_t prev_cand = 0xffffffffu;                           // candidate of the window before this round's first, and whether it was confirmed
    bool prev_ok = false;
    for (uint32_t o0 = 0; o0 < Lh; o0 += 64) {
        const uint32_t off = o0 + lane, x = ho + off;
        const bool in = off < Lh;
        const uint32_t cand = in ? b.canon[x] : 0xffffffffu;
        uint32_t before = hw::shfl(cand, (int)(lane ? lane - 1 : 0));
        if (lane == 0) before = prev_cand;
        const bool self = !in || cand == x;                                             // its own canonical form (or no window): nothing to confirm
        const bool cont = !self && off > 0 && before != x - 1 && cand == before + 1;    // continues the run of its predecessor (which is not its own canonical form)
        const uint32_t n = in ? window_len_at(b, off, Lh) : 0;
        // own evidence: a continuing window checks its last position (one cut by the haplotype's end adds none: it lies inside its predecessor); the head of a
        // run compares its whole window
        bool own = true;
        if (cont) { if (n == b.window_len) own = window_position_equal(b, x + n - 1, cand + n - 1); }
        else if (!self) own = window_bytes_equal(b, x, cand, n);
        // confirmed = nothing failed from the head of the run up to here; a run that came in from the round before carries that round's verdict
        const uint64_t bad = hw::ballot(!self && !own), starts = hw::ballot(!cont);
        const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        const uint64_t s_here = starts & upto;
        const uint32_t start = s_here ? 63u - (uint32_t)__builtin_clzll(s_here) : 0u;
        const uint64_t range = upto & ~((1ull << start) - 1ull);
        bool ok = (bad & range) == 0 && (s_here != 0 || prev_ok);
        if (cont && own && !ok) ok = window_bytes_equal(b, x, cand, n);                 // the run broke before this window (a colliding key upstream): on its own, then
        if (!self) b.canon[x] = ok ? cand : x;
        prev_cand = hw::shfl(cand, 63); prev_ok = hw::shfl((uint32_t)(!self && ok ? 1u : 0u), 63) != 0;
    }
}

// Match: one lane per read, 64 consecutive reads per wave (the pair arrays of a haplotype are read-major: coalesced), the haplotypes of the
// segment one after the other. Each read keeps the distinct pair hashes it has met (+ haplotype) in a small open-addressing table in LDS,
// column-major so that lanes never collide; a pair whose hash is in the table becomes a CANDIDATE for sharing (pair_rep), nothing else
// changes yet. Where a region continues in the next slice the table's entries are left in dd_* and re-inserted there.
constexpr uint32_t kDedupSlots = 64;                            // table slots per read (kDedupReps entries at most)
OCT_KERNEL(k_dedup_match)(DevBatch b, const DedupSeg* segs, uint32_t n_segs)
{
    OCT_DYN_SMEM(smem);
    uint32_t* tab_hash = (uint32_t*)smem;                       // [kDedupSlots][64], 0 = empty (pair hashes are never 0)
    uint16_t* tab_hap = (uint16_t*)(tab_hash + kDedupSlots * 64);   // [kDedupSlots][64] haplotype, counted from the region's first
    const uint32_t lane = hw::thread_idx() & 63, tile = hw::block_idx();
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (segs[mid].tile0 <= tile) lo = mid; else hi = mid; }
    const DedupSeg sg = segs[lo];
    const uint32_t rl = (tile - sg.tile0) * 64 + lane;          // read within the region
    if (rl >= sg.n_reads) return;
    const uint32_t r = sg.read0 + rl;
    for (uint32_t j = 0; j < kDedupSlots; ++j) tab_hash[j * 64 + lane] = 0;
    uint32_t n_seen = 0;
    // the haplotype that first showed `key`, or kNoPair after putting (key, hap) into the table (room permitting)
    auto find_or_insert = [&](uint32_t key, uint32_t hap) -> uint32_t {
        for (uint32_t slot = (key >> 8) % kDedupSlots; ; slot = (slot + 1) % kDedupSlots) {
            const uint32_t at = tab_hash[slot * 64 + lane];
            if (at == key) return sg.hap_first + tab_hap[slot * 64 + lane];
            if (at == 0) {
                if (n_seen < kDedupReps) { tab_hash[slot * 64 + lane] = key; tab_hap[slot * 64 + lane] = (uint16_t)(hap - sg.hap_first); ++n_seen; }
                return kNoPair;
            }
        }
    };
    if (sg.resumes) {                                           // what this read met among the region's haplotypes in the previous slice
        const uint32_t n = b.dd_n[r];
        for (uint32_t j = 0; j < n; ++j) find_or_insert(b.dd_hash[(size_t)j * b.n_reads + r], b.dd_hap[(size_t)j * b.n_reads + r]);
    }
    // Eight haplotypes' hashes per round, the NEXT round's loads in flight while this round's keys go through the table (a lone wave per SIMD hides no latency: a region
    // of 200 haplotypes is 25 dependent rounds, and on a device batch of 64 regions the launch waits for exactly that wave). Every haplotype of a region pairs with the
    // region's reads, so haplotype h's pairs start at pair_base + (h - hap_first) * n_reads: no look-up of hap_pair_off in front of each load.
    const uint64_t pair_base = b.hap_pair_off[sg.hap_first] + rl;
    auto pair_at = [&](uint32_t hp) -> uint64_t { return pair_base + (uint64_t)(hp - sg.hap_first) * sg.n_reads; };
    uint32_t keys[8], next[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) keys[u] = sg.hap_lo + u < sg.hap_hi ? b.pair_hash[pair_at(sg.hap_lo + u)] : 0u;
    for (uint32_t h0 = sg.hap_lo; h0 < sg.hap_hi; h0 += 8) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) next[u] = h0 + 8 + u < sg.hap_hi ? b.pair_hash[pair_at(h0 + 8 + u)] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            if (!keys[u]) continue;
            const uint32_t found = find_or_insert(keys[u], h0 + u);
            if (found != kNoPair) b.pair_rep[pair_at(h0 + u)] = found;                  // (the haplotype: k_dedup_verify turns it into the pair, or forgets it)
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) keys[u] = next[u];
    }
    if (sg.continues) {
        uint32_t n = 0;
        for (uint32_t slot = 0; slot < kDedupSlots; ++slot) {
            const uint32_t at = tab_hash[slot * 64 + lane];
            if (at) { b.dd_hash[(size_t)n * b.n_reads + r] = at; b.dd_hap[(size_t)n * b.n_reads + r] = sg.hap_first + tab_hap[slot * 64 + lane]; ++n; }
        }
        b.dd_n[r] = n;
    }
}

// Verify: one thread per pair. A candidate whose fast-path minimum, task classes, haplotype length, positions and canonical windows equal
// those of the pair it points at drops its DP tasks (classes, scan counts) and keeps the pointer; any other candidate forgets it.
// tile_sums != null: k_classify left RAW counts; this kernel stores them scanned across the workgroup once the shared pairs' counts are dropped (store_scanned_local)
OCT_KERNEL(k_dedup_verify)(DevBatch b, uint64_t pair0, uint64_t pair1, uint4* cnt, uint4* cnt_late, uint4* tile_sums, uint4* tile_sums_late)
{
    __shared__ uint4 sh_scan[4];
    const uint64_t e = pair0 + (uint64_t)hw::block_idx() * hw::block_dim() + hw::thread_idx();
    const uint64_t e_wave = wave_first_index(pair0);
    unsigned long long st_score = 0, st_trace = 0, st_cells = 0, st_pairs = 0;
    uint4 my_cnt = make_uint4(0, 0, 0, 0), my_late = make_uint4(0, 0, 0, 0);
    if (tile_sums && e < pair1) { my_cnt = cnt[e - pair0]; if (cnt_late) my_late = cnt_late[e - pair0]; }
    if (e < pair1) {
        const uint32_t h2 = b.pair_rep[e];                                     // the matcher's candidate: the HAPLOTYPE whose pair with this read looked the same
        if (h2 != kNoPair) {
            PairView v, w;
            const uint32_t h = upper_bound_near(b.hap_pair_off, b.n_haps + 1, e_wave, e);
            const uint64_t shared = b.hap_pair_off[h2] + (e - b.hap_pair_off[h]);
            // (fast-path minima from pair_fast: the other pair may lie in an earlier slice, whose DP results are already landing in pair_best)
            v.e = e; v.cls = b.pair_cls[e]; v.best = b.pair_fast[e]; v.ho = b.hoff[h]; v.Lh = b.hoff[h + 1] - v.ho;
            w.e = shared; w.cls = b.pair_cls[shared]; w.best = b.pair_fast[shared]; w.ho = b.hoff[h2]; w.Lh = b.hoff[h2 + 1] - w.ho;
            if (v.cls && pair_views_same(b, v, w)) {
                const uint32_t r = b.reg_read0[b.hap_region[h]] + (uint32_t)(e - b.hap_pair_off[h]), T = b.roff[r + 1] - b.roff[r];
                uint32_t n_score = 0, n_trace = 0;
                for (uint32_t slot = 0; slot <= (uint32_t)b.max_pos; ++slot) { const uint32_t kd = (v.cls >> (2 * slot)) & 3u; n_score += kd == 1u ? 1u : 0u; n_trace += kd >= 2u ? 1u : 0u; }
                st_score = n_score; st_trace = n_trace; st_cells = (unsigned long long)(n_score + n_trace) * 2ull * (uint32_t)b.band * (T + (uint32_t)b.band); st_pairs = 1;
                b.pair_cls[e] = 0;
                b.pair_rep[e] = (uint32_t)shared;                                     // from here on: the pair whose result this one reads
                my_cnt = make_uint4(0, 0, 0, 0); my_late = make_uint4(0, 0, 0, 0);
                if (!tile_sums) { cnt[e - pair0] = my_cnt; if (cnt_late) cnt_late[e - pair0] = my_late; }
            } else {
                b.pair_rep[e] = kNoPair;
            }
        }
    }
    if (tile_sums) {
        store_scanned_local(my_cnt, e, pair0, pair1, cnt, tile_sums, sh_scan);
        if (cnt_late) store_scanned_local(my_late, e, pair0, pair1, cnt_late, tile_sums_late, sh_scan);
    }
    st_score = wave_sum(st_score); st_trace = wave_sum(st_trace); st_cells = wave_sum(st_cells); st_pairs = wave_sum(st_pairs);
    if ((hw::thread_idx() & 63) == 0 && st_pairs) {
        unsigned long long* st = b.stats + (size_t)(hw::block_idx() % kStatSlots) * kStatStride;
        hw::atomic_add_u64(st + 8, st_score); hw::atomic_add_u64(st + 9, st_trace); hw::atomic_add_u64(st + 10, st_cells); hw::atomic_add_u64(st + 11, st_pairs);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of the per-pair task counts (uint4, one component per task kind) over n = n_pairs + 1 items: tile-local in the kernel that makes the counts, finished by k_scan_finish
// (rounds 1-4 scanned with k_scan_tiles x 3 + k_hap_bases per count array, or k_scan_bases for region-sized batches: retired in round 6, profiles/EXPERIMENTS.md)
// ------------------------------------------------------------------------------------------------------------------
OCT_DEVICE uint4 add4(uint4 a, uint4 b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
constexpr uint32_t kHapBaseThreads = 1024;
OCT_DEVICE uint4 shfl4(uint4 v, uint32_t src) { return make_uint4(hw::shfl(v.x, (int)src), hw::shfl(v.y, (int)src), hw::shfl(v.z, (int)src), hw::shfl(v.w, (int)src)); }
OCT_DEVICE uint4 sub4(uint4 a, uint4 b) { return make_uint4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
// exclusive prefix of `v` over the kHapBaseThreads threads of the workgroup; *total = the sum over all of them. sh: 16 words of LDS.
OCT_DEVICE uint4 block_scan_excl(uint4 v, uint4* sh, uint4* total)
{
    const uint32_t tid = hw::thread_idx(), lane = tid & 63u, wv = tid >> 6;
    uint4 inc = v;
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint4 o = shfl4(inc, lane >= d ? lane - d : lane); if (lane >= d) inc = add4(inc, o); }
    hw::block_sync();                                           // (an earlier round's reads of sh are over)
    if (lane == 63) sh[wv] = inc;
    hw::block_sync();
    uint4 before = make_uint4(0, 0, 0, 0), all = make_uint4(0, 0, 0, 0);
    for (uint32_t w = 0; w < kHapBaseThreads / 64; ++w) { const uint4 s = sh[w]; if (w < wv) before = add4(before, s); all = add4(all, s); }
    *total = all;
    return add4(before, sub4(inc, v));
}

// Any size, TWO launches (a region server's device batch is a chain of dependent launches):
//   * the kernel that makes the counts - k_classify, or k_dedup_verify where pairs are shared - scans them across its own workgroup (a "tile" of 256 pairs) and
//     stores every pair's tile-local exclusive prefix and the tile's total (block_scan_local);
//   * k_scan_finish, one workgroup per count array: tile totals -> tile prefixes (in place), per-haplotype bases and totals of both count arrays from local prefix + tile prefix.
// Readers of the counts (k_emit) add the tile prefix themselves: ScanView. (First form of the round: one launch whose last workgroup did the second step behind a
// device-scope counter - 55 us for 150 k pairs against 48 us for the eight launches it replaced, and 10 % on the 12.8 M-pair step: every workgroup's release fence is a
// write-back of its XCD's L2 on this chip, 392 of them per slice beside other slices' backpointer stores. profiles/EXPERIMENTS.md.)
struct ScanView { const uint4* cnt; const uint4* tile_pref; };       // tile_pref null: cnt is scanned globally
constexpr uint32_t kScanLocalShift = 8, kScanLocalTile = 1u << kScanLocalShift;      // = the 256 threads of k_classify / k_dedup_verify
OCT_DEVICE uint4 scanned(const ScanView& v, uint64_t i) { const uint4 c = v.cnt[i]; return v.tile_pref ? add4(c, v.tile_pref[i >> kScanLocalShift]) : c; }
OCT_KERNEL(k_scan_finish)(DevBatch b, uint32_t hap0, uint32_t hap1, const uint4* cnt0, const uint4* cnt1, uint64_t pair0, uint32_t n_tiles,
                          uint4* tile_sums0, uint4* tile_sums1, uint4* hap_base0, uint4* hap_base1, uint4* totals0, uint4* totals1, uint32_t group)
{
    OCT_DYN_SMEM(smem);
    uint4* sh = (uint4*)smem;                                   // [16]
    const uint32_t tid = hw::thread_idx();
    // one workgroup per count array (round 6: the two arrays' rounds of dependent loads ran one after the other in ONE workgroup - 0.73 ms on the 2,000-region stream in one slice)
    const uint32_t n_arr = hw::grid_dim(), a = hw::block_idx();
    {
        const uint4* cnt = a ? cnt1 : cnt0; uint4* tsum = a ? tile_sums1 : tile_sums0; uint4* hap_base = a ? hap_base1 : hap_base0;
        uint4 all_a;
        // tile totals -> exclusive tile prefixes, in place: eight consecutive tiles per thread and round (coalesced runs, one workgroup scan per 8,192 tiles = 2 M pairs)
        constexpr uint32_t PER = 8;
        uint4 carry = make_uint4(0, 0, 0, 0);
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += kHapBaseThreads * PER) {
            const uint32_t i0 = t0 + tid * PER;
            uint4 v[PER], sum = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j) { v[j] = i0 + j < n_tiles ? tsum[i0 + j] : make_uint4(0, 0, 0, 0); sum = add4(sum, v[j]); }
            uint4 round_total;
            uint4 run = add4(carry, block_scan_excl(sum, sh, &round_total));
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j) if (i0 + j < n_tiles) { tsum[i0 + j] = run; run = add4(run, v[j]); }
            carry = add4(carry, round_total);
        }
        hw::block_sync();                                       // the prefixes are read across threads below
        const ScanView view {cnt, tsum};
        // Per-haplotype bases: rounds of 4,096 consecutive haplotypes, four per thread - ONE pass: the five scanned counts at a thread's haplotype boundaries (coalesced offsets, ten
        // independent gathers in flight), the four padded counts, a workgroup scan per round, four stores. (The first form gave every thread a contiguous run of n / 1,024 haplotypes and walked it
        // twice - sum, then bases - recomputing the counts: 1.56 ms for the 2,000-region stream's 49 k haplotypes in one slice, twice k_hap_bases' time.)
        auto up = [&](uint32_t c) { return (c + group - 1) / group * group; };
        constexpr uint32_t PERH = 4;
        uint4 carry_h = make_uint4(0, 0, 0, 0);
        for (uint32_t h0 = hap0; h0 < hap1; h0 += kHapBaseThreads * PERH) {
            const uint32_t hb = h0 + tid * PERH;
            uint4 edge[PERH + 1];
#pragma unroll
            for (uint32_t u = 0; u <= PERH; ++u) edge[u] = hb + u <= hap1 ? scanned(view, b.hap_pair_off[hb + u] - pair0) : make_uint4(0, 0, 0, 0);
            uint4 v[PERH], sum = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t u = 0; u < PERH; ++u) {
                v[u] = hb + u < hap1 ? make_uint4(up(edge[u + 1].x - edge[u].x), up(edge[u + 1].y - edge[u].y), up(edge[u + 1].z - edge[u].z), up(edge[u + 1].w - edge[u].w)) : make_uint4(0, 0, 0, 0);
                sum = add4(sum, v[u]);
            }
            uint4 round_total;
            uint4 run = add4(carry_h, block_scan_excl(sum, sh, &round_total));
#pragma unroll
            for (uint32_t u = 0; u < PERH; ++u) if (hb + u < hap1) { hap_base[hb + u] = run; run = add4(run, v[u]); }
            carry_h = add4(carry_h, round_total);
        }
        all_a = carry_h;
        if (tid == 0) *(a ? totals1 : totals0) = all_a;
        // one traceback launch may take a flavour's traceback list AND its late-start list: together they must fit the scratch the host provisioned. The two workgroups
        // meet in one counter word (a spare slot of the statistics block, cleared with it): arrivals << 60 | generic tasks << 28 | fast tasks; whoever arrives last compares.
        if (tid == 0 && b.dsl_trace_cap) {
            const uint32_t lim = (1u << 28) - 1u;
            const uint32_t fast = a ? all_a.x : all_a.y, gen = a ? all_a.y : all_a.w;
            const unsigned long long mine = 1ull << 60 | (unsigned long long)(gen < lim ? gen : lim) << 28 | (unsigned long long)(fast < lim ? fast : lim);
            unsigned long long sum = mine;
            if (n_arr > 1) { const unsigned long long old = hw::atomic_add_u64(b.stats + kScanMeetSlot, mine); sum = (old >> 60) ? old + mine : 0ull; }
            if ((sum >> 60) == n_arr && ((uint32_t)(sum & lim) > b.dsl_trace_cap || (uint32_t)((sum >> 28) & lim) > b.dsl_trace_cap)) *b.dsl_overflow = 1ull;
        }
    }
}

struct TaskArrays { DevTask* t[kNumKinds]; };

// A region-sized call's inputs, from the handle's pinned staging buffer (mapped into the device) to their device block: a kernel starts sooner than a DMA copy
// (~5 us against ~12 us before the first byte moves), and ~100 KB over the host link are a few microseconds either way.
OCT_KERNEL(k_copy_from_host)(uint4* dst, const uint4* src, uint32_t n16)
{
    const uint32_t i = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (i < n16) dst[i] = src[i];
}

// Device-sized launches: first task and length of one of the six task lists, from k_hap_bases' totals in device memory (uniform: scalar loads)
// Physical order of the lists in the array: score-only fast, traceback fast, LATE fast, score-only generic, traceback generic, LATE generic - a flavour's
// late-start list lies right behind its traceback list, so that ONE traceback launch and ONE walk take both (ref.join_late; round 5: the second traceback
// launch of a region-sized step was a single round of workgroups on an otherwise idle chip, and its walk one more link in the chain).
OCT_DEVICE void task_list_range(const TaskListRef& ref, uint32_t& first, uint32_t& n, uint32_t* n_main = nullptr)   // n_main: the tasks of the list proper (behind them: the joined late-start list)
{
    const uint4 a = *ref.totals;
    const uint4 l = ref.totals_late ? *ref.totals_late : make_uint4(0, 0, 0, 0);
    // (scalar selects, no indexed arrays: those went to scratch memory in every kernel's prologue)
    const uint32_t sf = a.x, tf = a.y, lf = l.x, sg = a.z, tg = a.w, lg = l.y;       // in physical order
    const int id = ref.list;
    const bool join = ref.join_late != 0;
    first = id == 0 ? 0u : id == 1 ? sf : id == 4 ? sf + tf : id == 2 ? sf + tf + lf : id == 3 ? sf + tf + lf + sg : sf + tf + lf + sg + tg;
    const uint32_t own = id == 0 ? sf : id == 1 ? tf : id == 4 ? lf : id == 2 ? sg : id == 3 ? tg : lg;
    n = own + (join && id == 1 ? lf : join && id == 3 ? lg : 0u);
    if (n_main) *n_main = own;
    if (ref.overflow && *ref.overflow) { first = 0; n = 0; if (n_main) *n_main = 0; }
}

// Pass 2: write the DP tasks of every pair at hap_base + (scanned count - scanned count at the haplotype's first pair).
OCT_KERNEL(k_emit)(DevBatch b, uint64_t pair0, uint64_t pair1, const uint4* cnt, const uint4* hap_base, TaskArrays out,
                   const uint4* cnt_late, const uint4* hap_base_late, TaskArrays out_late, TaskListRef ref, uint32_t group,
                   const uint4* tile_pref, const uint4* tile_pref_late)      // tile prefixes of k_scan_fused's tile-local scans (null: the counts are scanned globally)
{
    const ScanView sv {cnt, tile_pref}, sv_late {cnt_late, tile_pref_late};
    const uint64_t e = pair0 + (uint64_t)hw::block_idx() * hw::block_dim() + hw::thread_idx();
    const uint64_t e_wave = wave_first_index(pair0);
    if (e >= pair1) return;
    if (ref.totals) {                                           // device-sized: the six lists one behind the other in out.t[0]
        if (ref.overflow && *ref.overflow) return;
        DevTask* base = out.t[0];
        for (int k = 0; k < 6; ++k) {
            TaskListRef q = ref; q.list = k;
            uint32_t first, n; task_list_range(q, first, n);
            if (k < kNumKinds) out.t[k] = base + first; else out_late.t[k - kNumKinds] = base + first;
        }
    }
    const uint32_t cls = b.pair_cls[e];
    if (!cls) return;
    const uint32_t h = upper_bound_near(b.hap_pair_off, b.n_haps + 1, e_wave, e);
    const uint32_t r = b.reg_read0[b.hap_region[h]] + (uint32_t)(e - b.hap_pair_off[h]);
    const bool generic = b.wide || !(b.racgt[r] && b.hclean[h]);
    const uint4 s = scanned(sv, e - pair0), s0 = scanned(sv, b.hap_pair_off[h] - pair0), z = scanned(sv, b.hap_pair_off[h + 1] - pair0), hb = hap_base[h];
    uint32_t at_score = generic ? hb.z + (s.z - s0.z) : hb.x + (s.x - s0.x);
    uint32_t at_trace = generic ? hb.w + (s.w - s0.w) : hb.y + (s.y - s0.y);
    // one past the haplotype's last real task of each list: whoever writes the task before it also writes the padding copies behind it
    const uint32_t end_score = generic ? hb.z + (z.z - s0.z) : hb.x + (z.x - s0.x), end_trace = generic ? hb.w + (z.w - s0.w) : hb.y + (z.y - s0.y);
    DevTask* ts = out.t[generic ? kScoreGen : kScoreFast]; DevTask* tt = out.t[generic ? kTraceGen : kTraceFast];
    uint32_t at_late = 0, end_late = 0; DevTask* tl = nullptr;
    if (cnt_late) {
        const uint4 q = scanned(sv_late, e - pair0), q0 = scanned(sv_late, b.hap_pair_off[h] - pair0), qz = scanned(sv_late, b.hap_pair_off[h + 1] - pair0), qb = hap_base_late[h];
        at_late = generic ? qb.y + (q.y - q0.y) : qb.x + (q.x - q0.x); tl = out_late.t[generic ? 1 : 0];
        end_late = generic ? qb.y + (qz.y - q0.y) : qb.x + (qz.x - q0.x);
    }
    const uint32_t* P = b.pos + e * (uint64_t)b.max_pos;
    const uint32_t B = (uint32_t)b.band;
    // Every haplotype's run in a list is padded to whole task groups (a DP task group never straddles two haplotypes): copies of its last task with
    // pair = kPadTask. hap_base already counts them (k_hap_bases).
    auto put = [&](DevTask* list, uint32_t& at, uint32_t end, uint32_t first, const DevTask& t) {
        list[at++] = t;
        if (at != end) return;
        const uint32_t n = end - first, padded = (n + group - 1) / group * group;
        DevTask q = t; q.pair = kPadTask;
        for (uint32_t i = n; i < padded; ++i) list[first + i] = q;
    };
    for (uint32_t slot = 0; slot <= (uint32_t)b.max_pos; ++slot) {
        const uint32_t k = (cls >> (2 * slot)) & 3u;
        if (!k) continue;
        const uint32_t p = slot < (uint32_t)b.max_pos ? P[slot] : b.pair_extra[e];
        DevTask t; t.pair = (uint32_t)e; t.read = r; t.hap = h; t.off = p > B ? p - B : 0;
        if (k == 1) put(ts, at_score, end_score, generic ? hb.z : hb.x, t);
        else if (k == 2) put(tt, at_trace, end_trace, generic ? hb.w : hb.y, t);
        else put(tl, at_late, end_late, generic ? hap_base_late[h].y : hap_base_late[h].x, t);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Window pairing (round 6). k_dp packs two tasks into the halves of every lane; when both read the SAME haplotype window (same haplotype, offset and strand) the
// per-column gap words arrive packed for both halves and two v_perm per iteration go (dp_groups, PAIRED). Big batches offer such pairs in plenty - 100 k reads over
// ~240 (offset, strand) classes per haplotype - but k_emit writes a haplotype's tasks in read order. This kernel re-orders one haplotype's run of one task list
// (out of place): a counting sort by key = offset * 2 + strand in LDS; every class gives its even part to the front of the run, class after class (tasks 2i, 2i + 1
// of the run then share a window), its odd one out goes behind them; paired_end[hap] = list index where the front part ends. Which task of a class meets which is
// decided by the order the LDS atomics land in - every task's result is its own, so the matrix does not depend on it.
// Grid: one workgroup per (haplotype of the slice, list); runs = {first task, one past the last} of every such run in its list.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPairSortThreads = 1024, kPairSortMaxLh = 2040;       // keys = 2 x (longest haplotype) + 2: three arrays of them stay below 64 KB of LDS
OCT_HD uint32_t pair_sort_keys(uint32_t lh_cap) { return (2 * lh_cap + 2 + 3) & ~3u; }
inline uint32_t pair_sort_lds_bytes(uint32_t lh_cap) { return 3 * pair_sort_keys(lh_cap) * 4 + 16 * 16; }
// one task list of a slice: `comp` picks the list's component (0 x .. 3 w) of the per-haplotype bases k_scan_finish left (list index of a haplotype's first task); total = the list's length
struct PairSortList { const DevTask* in; DevTask* out; uint32_t* paired_end; const uint4* hap_base; uint32_t comp, total; };
OCT_DEVICE uint32_t comp4(const uint4& v, uint32_t c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
OCT_MAX_THREADS(1024) OCT_KERNEL(k_pair_sort)(PairSortList l0, PairSortList l1, PairSortList l2, const uint8_t* rrev, uint32_t hap0, uint32_t n_haps_slice, uint32_t n_keys)
{
    OCT_DYN_SMEM(smem);
    uint32_t* cnt = (uint32_t*)smem;                 // [n_keys] class sizes, then the classes' cursors
    uint32_t* front = cnt + n_keys;                  // [n_keys] where a class's even part starts (relative to the run)
    uint32_t* back = front + n_keys;                 // [n_keys] where its odd one out goes
    uint4* sh = (uint4*)(back + n_keys);             // [16] scan scratch
    const uint32_t which = hw::block_idx() / n_haps_slice, hs = hw::block_idx() % n_haps_slice, tid = hw::thread_idx();
    const PairSortList l = which == 0 ? l0 : which == 1 ? l1 : l2;
    if (!l.in) return;
    const uint32_t t0 = comp4(l.hap_base[hap0 + hs], l.comp), t1 = hs + 1 < n_haps_slice ? comp4(l.hap_base[hap0 + hs + 1], l.comp) : l.total, n = t1 - t0;
    for (uint32_t k = tid; k < n_keys; k += kPairSortThreads) cnt[k] = 0;
    hw::block_sync();
    auto key_of = [&](const DevTask& t) -> uint32_t { const uint32_t k = t.off * 2u + (rrev[t.read] ? 1u : 0u); return k < n_keys ? k : n_keys - 1; };
    // (four tasks per thread and trip, loads first: the passes are chains of dependent round trips - task, its read's strand, the counter - and a run of 40 k tasks is 40 trips deep)
    constexpr uint32_t U = 4;
    for (uint32_t i0 = tid; i0 < n; i0 += U * kPairSortThreads) {
        DevTask t[U]; uint32_t k[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { const uint32_t i = i0 + u * kPairSortThreads; t[u] = l.in[t0 + (i < n ? i : 0)]; }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) k[u] = key_of(t[u]);
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) if (i0 + u * kPairSortThreads < n) hw::atomic_add_lds_u32(&cnt[k[u]], 1u);
    }
    hw::block_sync();
    // exclusive scans of the even parts and of the odd ones out over the classes: a contiguous stretch of classes per thread, one workgroup scan
    const uint32_t per = (n_keys + kPairSortThreads - 1) / kPairSortThreads, k0 = tid * per, k1 = k0 + per < n_keys ? k0 + per : n_keys;
    uint4 mine = make_uint4(0, 0, 0, 0);
    for (uint32_t k = k0; k < k1; ++k) { mine.x += cnt[k] & ~1u; mine.y += cnt[k] & 1u; }
    uint4 total;
    const uint4 before = block_scan_excl(mine, sh, &total);
    uint32_t run_f = before.x, run_b = total.x + before.y;
    for (uint32_t k = k0; k < k1; ++k) { const uint32_t c = cnt[k]; front[k] = run_f; back[k] = run_b; run_f += c & ~1u; run_b += c & 1u; }
    hw::block_sync();
    for (uint32_t k = tid; k < n_keys; k += kPairSortThreads) cnt[k] &= ~1u;                 // a class's even part: cursors count up from it ... (see below)
    hw::block_sync();
    // second pass: a task takes the next place of its class's even part while there is one (an LDS counter per class counts DOWN from the even size), else the odd place
    for (uint32_t i0 = tid; i0 < n; i0 += U * kPairSortThreads) {
        DevTask t[U]; uint32_t k[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { const uint32_t i = i0 + u * kPairSortThreads; t[u] = l.in[t0 + (i < n ? i : 0)]; }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) k[u] = key_of(t[u]);
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) if (i0 + u * kPairSortThreads < n) {
            const uint32_t left = hw::atomic_add_lds_u32(&cnt[k[u]], 0xffffffffu);            // returns the value before the decrement
            const uint32_t dst = (left != 0 && left <= 0x7fffffffu) ? front[k[u]] + (left - 1) : back[k[u]];
            l.out[t0 + dst] = t[u];
        }
    }
    if (tid == 0) l.paired_end[hap0 + hs] = t0 + total.x;
}

// ------------------------------------------------------------------------------------------------------------------
// the banded min-plus DP
// ------------------------------------------------------------------------------------------------------------------
// Both shifts are ONE v_or_b32_dpp (B = 16, 64) or v_and_b32_dpp + v_or_b32 (B = 8, 32): the DPP move zeroes the lane that has no source
// (bound_ctrl), and the fill arrives through a loop-invariant lane constant instead of a v_mov of the fill before every shift.
template <int B> OCT_DEVICE uint32_t shift_up(uint32_t fill, uint32_t v, uint32_t li)     // lane i <- lane i-1, first lane of a task row <- fill
{
    const uint32_t fw = li == 0 ? fill : 0u;
    if constexpr (B == 16) return hw::dpp_row_shr1_z(v) | fw;
    else if constexpr (B == 64) return hw::dpp_wave_shr1_z(v) | fw;
    else if constexpr (B == 8) return (hw::dpp_row_shr1_z(v) & (li == 0 ? 0u : ~0u)) | fw;
    else return (hw::dpp_wave_shr1_z(v) & (li == 0 ? 0u : ~0u)) | fw;
}
template <int B> OCT_DEVICE uint32_t shift_down(uint32_t fill, uint32_t v, uint32_t li)   // lane i <- lane i+1, last lane of a task row <- fill
{
    const uint32_t fw = li == B - 1 ? fill : 0u;
    if constexpr (B == 16) return hw::dpp_row_shl1_z(v) | fw;
    else if constexpr (B == 64) return hw::dpp_wave_shl1_z(v) | fw;
    else if constexpr (B == 8) return (hw::dpp_row_shl1_z(v) & (li == B - 1 ? 0u : ~0u)) | fw;
    else return (hw::dpp_wave_shl1_z(v) & (li == B - 1 ? 0u : ~0u)) | fw;
}

template <bool V> struct BoolC { static constexpr bool value = V; };

// LDS footprint of one DP workgroup (bytes) — must match the carve-up in k_dp.
OCT_HD constexpr uint32_t dp_rec_n(uint32_t t_cap, uint32_t B) { return (t_cap + 2 * B + 12) & ~3u; }
#ifndef OCT_TILE_STRIDE
#define OCT_TILE_STRIDE 66             // (a build-time knob for A/B libraries: tools/build_variant.sh)
#endif
constexpr uint32_t kTileStride = OCT_TILE_STRIDE;   // dwords per row of the 16 x 64 backpointer transpose tile (66: conflict-free both ways)
// rec_chunk (k_dp / k_dp_pair only): 0 = a wave stages its reads' records whole (index j = read position j - B, T + 2B + 12 entries per row); n = it stages n iterations'
// worth at a time (n + 2B + 12 entries) and restages every n iterations. 500-base reads (the chunks the reference's PacBio configuration cuts long reads into)
// against 1.8 kb haplotypes would otherwise take 115 KB of LDS per workgroup = one wave per SIMD.
OCT_HD constexpr uint32_t dp_rec_rows_n(uint32_t t_cap, uint32_t B, uint32_t rec_chunk) { return rec_chunk ? dp_rec_n(rec_chunk, B) : dp_rec_n(t_cap, B); }
inline uint32_t dp_lds_bytes(uint32_t t_cap, uint32_t lh_cap, uint32_t B, bool trace, uint32_t rec_chunk = 0, bool paired = false)   // paired: room for the 24-byte columns of window-paired segments
{
    const uint32_t rows = 64 / B;
    return ((lh_cap + 8 + 1) & ~1u) * (paired ? 24 : 16) + kBlockWaves * rows * dp_rec_rows_n(t_cap, B, rec_chunk) * 8 + (trace ? kBlockWaves * 16 * kTileStride * 4 : 0);
}
// traceback scratch: per task group, ceil(iterations / 16) tiles of 64 lanes x 16 iterations, each lane's 16 dwords contiguous
OCT_HD constexpr uint32_t bp_tiles(uint32_t t_cap, uint32_t B) { return (t_cap + B + 15) / 16 + 1; }

// OCT_DP_ABLATE=n (A/B builds only, tools/build_variant.sh; results are NOT the product's): the kernel without one of its phases, to price that phase by difference -
// 1 haplotype-table staging, 2 read-record staging, 3 end-cell capture + the final reduction, 4 the tile flush to HBM (traceback), 5 tile writes and flush (traceback)
#ifndef OCT_DP_ABLATE
#define OCT_DP_ABLATE 0
#endif
// (the body of k_dp: workgroup `blk` of `nblk` - a launch may hold the workgroups of two task lists, see k_dp_pair)
template <int B, bool TRACE, bool GENERIC, bool FASTADD>
OCT_DEVICE void dp_groups(const DpParams& p, const uint32_t blk, const uint32_t nblk)
{
    constexpr uint32_t ROWS = 64 / B, G = 2 * ROWS;
    OCT_DYN_SMEM(smem);
    const uint32_t tid = hw::thread_idx(), lane = tid & 63, wave = hw::readfirstlane(tid >> 6);   // uniform, and known to be: group index and tile addresses live in SGPRs
    const uint32_t row = lane / B, li = lane % B;
    const uint32_t lh_n = (p.lh_cap + 8 + 1) & ~1u, rec_n = dp_rec_rows_n(p.t_cap, B, p.rec_chunk);
    uint2* tabF = (uint2*)smem;                      // [lh_n] forward-strand table of the current haplotype
    uint2* tabR = tabF + lh_n;                       // [lh_n] reverse-strand table
    // Window-paired segments (p.paired_end, round 6): the two tasks packed in a lane's halves read ONE haplotype window, so a column's gap words arrive packed for both
    // halves - no v_perm per iteration to bring two windows' words together - and the insertion's words carry nuc_prior already: per column ONE 16-byte entry of gap
    // words {gap open[x] x 2, gap extend[x] x 2, gap open[x - 1] x 2 + nuc, gap extend[x - 1] x 2 + nuc} (the same on both strands) and a 4-byte cap row per strand:
    // 24 bytes against the 16 of the two 8-byte tables - the traceback form keeps its three workgroups per CU (with 20 bytes per column AND strand it had two: + 3 % instead of - 6 %).
    constexpr bool CAN_PAIR = !GENERIC && FASTADD;
    uint4* tab4 = (uint4*)smem;
    uint32_t* tab1F = (uint32_t*)(tab4 + lh_n); uint32_t* tab1R = tab1F + lh_n;
    uint2* recs = (CAN_PAIR && p.paired_end) ? (uint2*)(tab1R + lh_n) : tabR + lh_n;   // [kBlockWaves][ROWS][rec_n] read-side records (of the whole reads, or of p.rec_chunk iterations at a time)
    uint32_t* tiles = (uint32_t*)(recs + kBlockWaves * ROWS * rec_n);   // [kBlockWaves][16][kTileStride] backpointer transpose tiles (TRACE)
    uint2* rec_row = recs + (wave * ROWS + row) * rec_n;
    uint32_t* tile = tiles + wave * 16 * kTileStride;

    const DevTask* tasks = p.tasks; uint32_t n_tasks = p.n_tasks;
    uint32_t late_from = p.late_from;                   // first task of the launch that may start its traceback late (the joined late-start list; 0: a late-start list on its own)
    if (p.ref.totals) { uint32_t first, n_main; task_list_range(p.ref, first, n_tasks, &n_main); tasks += first; late_from = p.ref.join_late ? n_main : p.late_from; }   // device-sized launch: this kernel's list, from device memory
    const uint32_t n_groups = n_tasks / G;
    const uint32_t NUC = p.nuc4;
    // State words are kept biased by 0x8000 per half (null_score_ -> 0x0000, infinity_ -> 0xF800): a wrapping add is bit-identical to the
    // reference's int16 add, the signed min becomes an unsigned min, and a half only carries into its neighbour where the reference's
    // own lane would wrap. FASTADD (chosen by the host when its bounds prove no lane can wrap) then uses plain
    // v_add_u32 (2 cycles) instead of v_pk_add_u16 (4 cycles).
    constexpr uint32_t INFB = INF2 ^ 0x80008000u, NULB = NUL2 ^ 0x80008000u;
    auto sadd = [](uint32_t x, uint32_t y) -> uint32_t { if constexpr (FASTADD) return x + y; else return hw::pk_add(x, y); };

    // (a host-sized launch has one workgroup per run of groups_per_block groups; a device-sized one a grid from the host's bound, strided)
    for (uint32_t g_begin = blk * p.groups_per_block; g_begin < n_groups; g_begin += nblk * p.groups_per_block) {
    const uint32_t g_end = g_begin + p.groups_per_block < n_groups ? g_begin + p.groups_per_block : n_groups;
    uint32_t seg = g_begin;
    while (seg < g_end) {
        // ---- haplotype segment [seg, seg_end): stage its two strand tables in LDS ----
        const uint32_t hap = tasks[seg * G].hap;
        uint32_t seg_end = seg + 1;
        while (seg_end < g_end && tasks[seg_end * G].hap == hap) ++seg_end;
        const uint32_t ho = p.hoff[hap], Lh = p.hoff[hap + 1] - ho;
        bool paired = false;
        if constexpr (CAN_PAIR) if (p.paired_end) {
            // the tasks of this haplotype before list index paired_end[hap] lie two by two on one window (k_pair_sort); the segment ends where they do
            const uint32_t pe = p.paired_end[hap], pg = pe > p.task0 ? (pe - p.task0) / G : 0u;
            if (seg < pg) { paired = true; if (seg_end > pg) seg_end = pg; }
        }
        hw::block_sync();
        if (OCT_DP_ABLATE == 1) {}
        else if (paired) {
            for (uint32_t x = tid; x < lh_n; x += kBlockWaves * 64) {
                const bool in = x < Lh, inp = x >= 1 && x - 1 < Lh;
                const uint2 f = in ? p.tabF[ho + x] : make_uint2(0, 0), r = in ? p.tabR[ho + x] : make_uint2(0, 0);
                const uint32_t gp = inp ? p.tabF[ho + x - 1].y : 0u;                          // (the gap words are the same on both strands)
                const uint32_t god = (f.y & 0xffffu) * 0x10001u, ged = (f.y >> 16) * 0x10001u;
                const uint32_t goi = (gp & 0xffffu) * 0x10001u + NUC, gei = (gp >> 16) * 0x10001u + NUC;
                tab4[x] = make_uint4(god, ged, goi, gei);
                tab1F[x] = f.x; tab1R[x] = r.x;
            }
        } else
        for (uint32_t x = tid; x < lh_n; x += kBlockWaves * 64) {
            const bool in = x < Lh;
            tabF[x] = in ? p.tabF[ho + x] : make_uint2(0, 0);
            tabR[x] = in ? p.tabR[ho + x] : make_uint2(0, 0);
        }
        hw::block_sync();

        auto run_group = [&](auto paired_c, const uint32_t g) {
            constexpr bool PAIRED = decltype(paired_c)::value;
            const DevTask tA = tasks[g * G + 2 * row], tB = tasks[g * G + 2 * row + 1];
            const uint32_t roA = p.roff[tA.read], TA = p.roff[tA.read + 1] - roA;
            const uint32_t roB = p.roff[tB.read], TB = p.roff[tB.read + 1] - roB;
            uint32_t Tmax = TA > TB ? TA : TB, Tmin = TA < TB ? TA : TB;
            for (int m = B; m < 64; m <<= 1) {
                const uint32_t o = hw::shfl_xor(Tmax, m), u = hw::shfl_xor(Tmin, m);
                Tmax = o > Tmax ? o : Tmax; Tmin = u < Tmin ? u : Tmin;
            }
            Tmax = hw::readfirstlane(Tmax); Tmin = hw::readfirstlane(Tmin);          // wave-uniform loop bounds (SGPRs)
            const uint32_t K4 = (Tmax + B + 3) & ~3u;                                 // iterations, run in quads

            // ---- stage the read-side records of this row: entry j holds read position t = k_base + j - B; 4 positions per lane per trip. With p.rec_chunk the row
            // holds the entries of the next rec_chunk iterations only (iteration k reads entries k - k_base + B - li and one beyond) and is restaged at every chunk border ----
            const uint32_t n_entries_all = K4 + B + 1;
            auto stage = [&](const uint32_t k_base) {
            if (OCT_DP_ABLATE == 2) return;
            const uint32_t n_need = p.rec_chunk && k_base + p.rec_chunk + B + 2 < n_entries_all ? p.rec_chunk + B + 2 : n_entries_all - k_base;
            hw::wave_lds_fence();                                                            // (a restage: every lane has read what it needed of the old entries)
            if constexpr (!GENERIC) {
                // fast cost: the two reads' precomputed rows (read_record_thread), interleaved into {selA, selB + 4, qA, qB}
                const uint4* rowA = (const uint4*)(p.rrec + (size_t)tA.read * p.rrec_stride) + (k_base >> 2);
                const uint4* rowB = (const uint4*)(p.rrec + (size_t)tB.read * p.rrec_stride) + (k_base >> 2);
                for (uint32_t j0 = 4 * li; j0 < n_need; j0 += 4 * B) {
                    const uint4 a = rowA[j0 >> 2], c = rowB[j0 >> 2];
                    rec_row[j0 + 0] = make_uint2(hw::perm(c.x, a.x, 0x05040100u) | 0x00040000u, hw::perm(c.x, a.x, 0x07060302u));
                    rec_row[j0 + 1] = make_uint2(hw::perm(c.y, a.y, 0x05040100u) | 0x00040000u, hw::perm(c.y, a.y, 0x07060302u));
                    rec_row[j0 + 2] = make_uint2(hw::perm(c.z, a.z, 0x05040100u) | 0x00040000u, hw::perm(c.z, a.z, 0x07060302u));
                    rec_row[j0 + 3] = make_uint2(hw::perm(c.w, a.w, 0x05040100u) | 0x00040000u, hw::perm(c.w, a.w, 0x07060302u));
                }
            } else
            for (uint32_t j0 = 4 * li; j0 < n_need; j0 += 4 * B) {
                const int32_t t0 = (int32_t)(k_base + j0) - B;
                auto load4 = [&](const uint8_t* base, uint32_t T) -> uint32_t {
                    if (t0 >= 0 && (uint32_t)t0 + 4 <= T) { uint32_t v; __builtin_memcpy(&v, base + t0, 4); return v; }
                    uint32_t v = 0;
                    for (int c = 0; c < 4; ++c) { const int32_t t = t0 + c; if (t >= 0 && (uint32_t)t < T) v |= (uint32_t)base[t] << (8 * c); }
                    return v;
                };
                const uint32_t wrA = load4(p.rbases + roA, TA), wqA = load4(p.rquals + roA, TA);
                const uint32_t wrB = load4(p.rbases + roB, TB), wqB = load4(p.rquals + roB, TB);
                for (int c = 0; c < 4; ++c) {
                    const int32_t t = t0 + c;
                    const bool inA = t >= 0 && (uint32_t)t < TA, inB = t >= 0 && (uint32_t)t < TB;
                    const uint32_t rA = (wrA >> (8 * c)) & 0xffu, rB = (wrB >> (8 * c)) & 0xffu;
                    const uint32_t qA = inA ? (wqA >> (8 * c)) & 0xffu : 64u, qB = inB ? (wqB >> (8 * c)) & 0xffu : 64u;   // max_quality_score_, :60,260,280
                    uint2 rec;
                    if constexpr (GENERIC) {
                        // {target char as int16 (0x7800 = the never-matching _inf fill before the read, '0' after it, :259,279), quality << 2}
                        const uint32_t cA = inA ? rA : (t < 0 ? 0x7800u : (uint32_t)'0'), cB = inB ? rB : (t < 0 ? 0x7800u : (uint32_t)'0');
                        rec = make_uint2(cA | cB << 16, (qA << 2) | (qB << 18));
                    } else {
                        // {v_perm selector picking this base's cap byte out of {capsB, capsA} (0x0d = 0xff = no cap), quality}
                        const uint32_t sA = inA ? base_code(rA) : 0x0du, sB = inB ? 4u + base_code(rB) : 0x0du;
                        rec = make_uint2(sA | 0x0c00u | sB << 16 | 0x0c000000u, qA | qB << 16);
                    }
                    rec_row[j0 + c] = rec;
                }
            }
            hw::wave_lds_fence();
            };
            stage(0);

            const uint2* pA = (p.rrev[tA.read] ? tabR : tabF) + tA.off + li;
            const uint2* pB = (p.rrev[tB.read] ? tabR : tabF) + tB.off + li;
            const uint4* q4 = tab4 + tA.off + li;                                      // PAIRED: the one window of both tasks - its gap words ...
            const uint32_t* q1 = (p.rrev[tA.read] ? tab1R : tab1F) + tA.off + li;      // ... and its strand's cap rows
            const uint2* rp = rec_row + (B - li);                      // rp[k] = the record of iteration k; rebased when the row is restaged (p.rec_chunk)
            const uint32_t kendA = TA + li, kendB = TB + li;          // the iteration whose M cells are this lane's end cells (t == T)
            uint4* bpg = TRACE ? (uint4*)(p.bp + (size_t)g * p.k_cap * 1024) : nullptr;   // this group's tiles (k_cap tiles of 4 KB)

            uint32_t M1 = INFB, I1 = INFB, D1 = INFB, M2 = INFB, I2 = INFB, D2 = INFB;     // :267
            uint32_t y1 = INFB;                                                             // min(M1, I1) carried between iterations
            uint32_t bestE = INFB, bestO = INFB;                                            // minscore :269, per diagonal parity
            // software pipeline: operands of iteration k are in registers when it starts, those of k+1 are in flight
            uint2 rr = rp[0];
            uint2 cA = make_uint2(0, 0), cB = cA, nA = cA, nB = cA;
            uint32_t GO = 0, GE = 0, GOn = 0, GEn = 0;
            if constexpr (!PAIRED) {
                cA = pA[0]; cB = pB[0]; nA = pA[1]; nB = pB[1];
                GO = hw::perm(cB.y, cA.y, 0x05040100u); GE = hw::perm(cB.y, cA.y, 0x07060302u);
                GOn = hw::perm(nB.y, nA.y, 0x05040100u); GEn = hw::perm(nB.y, nA.y, 0x07060302u);
            } else {
                // cA.x / nA.x = the cap rows of columns x and x + 1; GOn / GEn = the deletion's words of column x + 1; GO / GE = the insertion's words of column x (nuc_prior added)
                const uint4 e1 = q4[1];
                cA.x = q1[0]; nA.x = q1[1]; GOn = e1.x; GEn = e1.y; GO = e1.z; GE = e1.w;
            }

            // returns the packed match cost; fsrc = a word that is non-zero per half exactly where the walk would charge a flank penalty
            auto cost = [&](const uint2 r2, const uint2 a2, const uint2 b2, uint32_t& fsrc) -> uint32_t {
                const uint32_t a = a2.x, b = PAIRED ? a2.x : b2.x;                          // cost words of the two packed tasks (PAIRED: one window)
                if constexpr (GENERIC) {
                    // update_match_state with the reference's equality tests on raw bytes (:121-132)
                    const uint32_t hh = hw::perm(b, a, 0x0c040c00u), mm = hw::perm(b, a, 0x0c050c01u);
                    const uint32_t pp = hw::pk_shl2(hw::perm(b, a, 0x0c060c02u)), nn = hw::perm(b, a, 0x0c070c03u);
                    const uint32_t ne = hw::pk_min_u(r2.x ^ hh, 0x00010001u), nf = hw::pk_min_u(r2.x ^ mm, 0x00010001u);
                    const uint32_t inner = hw::pk_mad(nf, hw::pk_sub(r2.y, pp), pp);        // target == mask ? prior : quality
                    uint32_t c = hw::pk_mul(ne, hw::pk_min_i(r2.y, inner));                 // 0 where target == truth
                    const uint32_t nq = hw::pk_mad(nn, 0x88088808u, INF2);                  // truth == 'N' ? n_score_ (8) : infinity_
                    fsrc = ne;                                                              // target != truth (also charges 2 on 'N', even at quality 0)
                    return hw::pk_min_i(c, nq);
                } else {
                    const uint32_t cp = hw::perm(b, a, r2.x);                               // {capA, capB} for this read base
                    const uint32_t c = hw::pk_min_u(cp, r2.y);                              // min(quality, cap), unshifted
                    fsrc = c;                                                               // the flank penalty of this column is exactly c
                    return c;
                }
            };
            // bit `bit` (and bit + 16) set where the half's cost is non-zero; costs stay below 0x4000 so the add cannot leave its half
            auto flag_of = [](uint32_t c, int bit) -> uint32_t { const uint32_t msk = 0x00010001u << bit; return (c + (msk - 0x00010001u)) & msk; };
            auto add_cost = [&](uint32_t m, uint32_t c) -> uint32_t {
                if constexpr (GENERIC) return sadd(m, c); else return hw::pk_mad(c, 0x00040004u, m);   // + (c << trace_bits_)
            };

            // every 16 iterations: transpose the 16 x 64 tile through LDS so that each lane's 16 words become one 64-byte line and
            // the wave stores 4 fully coalesced 1 KB pieces (k_walk then fetches one line per 16 walk steps)
            auto flush_tile = [&](uint32_t kt) {
                if (OCT_DP_ABLATE >= 4) return;
                hw::wave_lds_fence();
                uint4* dst = bpg + (size_t)kt * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t src = 16 * j + (lane >> 2), kq = (lane & 3) * 4;
                    // (streaming stores: the tiles are read once, by the walk kernel, a launch later - 17 GB per launch that need not displace what the other slices' kernels keep in L2;
                    // 12.8 M-pair step 26.0 -> 25.3 ms, traceback launch 7.59 -> 7.40 ms: profiles/r06_s12_streaming_tile_stores.txt)
                    hw::store_streaming_u4(dst + j * 64 + lane, make_uint4(tile[(kq + 0) * kTileStride + src], tile[(kq + 1) * kTileStride + src],
                                                                             tile[(kq + 2) * kTileStride + src], tile[(kq + 3) * kTileStride + src]));
                }
                hw::wave_lds_fence();
            };

            auto quad = [&](uint32_t k0, auto init_c, auto cap_c, auto tr_c) {
                constexpr bool INIT = decltype(init_c)::value, CAP = decltype(cap_c)::value && OCT_DP_ABLATE != 3, TR = TRACE && decltype(tr_c)::value;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + u;
                    const uint2 rr_nx = rp[k + 1];                                          // prefetch iteration k+1
                    uint2 nnA = make_uint2(0, 0), nnB = nnA; uint4 e4 = make_uint4(0, 0, 0, 0); uint32_t e1 = 0;
                    if constexpr (PAIRED) { e4 = q4[k + 2]; e1 = q1[k + 2]; } else { nnA = pA[k + 2]; nnB = pB[k + 2]; }
                    uint32_t gate = 0;                                                      // 0 on an end cell, saturating +65535 elsewhere
                    if constexpr (CAP) gate = (k == kendA ? 0u : 0xffffu) | (k == kendB ? 0u : 0xffff0000u);
                    // ---- even diagonal s = 2k: lane li is cell (t = k-li, x = k+li) ----
                    uint32_t m1 = hw::pk_min_u(y1, D1);                                     // :284
                    if constexpr (INIT) { const bool first = k == li; m1 = first ? NULB : m1; M2 = first ? NULB : M2; }   // :282-283
                    if constexpr (CAP) bestE = hw::pk_min_u(bestE, hw::pk_add_sat_u(m1, gate));   // :285-291
                    uint32_t fe = 0, fo = 0;
                    const uint32_t ce = cost(rr, cA, cB, fe);
                    M1 = add_cost(m1, ce);                                                  // :292
                    const uint32_t x2 = hw::pk_min_u(M2, I2);
                    const uint32_t dsh = hw::pk_min_u(sadd(D2, GEn), sadd(x2, GOn));
                    D1 = shift_up<B>(INFB, dsh, li);                                        // :293-294
                    if constexpr (PAIRED) I1 = hw::pk_min_u(sadd(I2, GE), sadd(M2, GO));     // (nuc_prior rides in the words)
                    else I1 = sadd(hw::pk_min_u(sadd(I2, GE), sadd(M2, GO)), NUC);           // :295
                    uint32_t bpe = 0;
                    if constexpr (TR) {                                                     // update_traceback :147-163
                        const uint32_t tm = M1 & 0x00030003u, ti = I1 & 0x00030003u, td = D1 & 0x00030003u;
                        M1 ^= tm; I1 |= 0x00010001u; D1 |= 0x00030003u;   // I inherits label 0 or 1 only (its predecessors are M and I, all addends are multiples of 4): setting bit 0 relabels it
                        bpe = hw_lshl_or(td, 4, hw_lshl_or(ti, 2, tm | flag_of(fe, 15)));   // + "this match cell costs something" flag
                    }
                    // ---- odd diagonal s = 2k+1: lane li is cell (t, x+1) ----
                    const uint32_t m2 = hw::pk_min_u(x2, D2);                               // :308
                    if constexpr (CAP) bestO = hw::pk_min_u(bestO, hw::pk_add_sat_u(m2, gate));   // :309-315
                    const uint32_t co = cost(rr, nA, nB, fo);
                    M2 = add_cost(m2, co);                                                  // :316
                    y1 = hw::pk_min_u(M1, I1);
                    D2 = hw::pk_min_u(sadd(D1, GEn), sadd(y1, GOn));                        // :317
                    const uint32_t ish = PAIRED ? hw::pk_min_u(sadd(I1, GE), sadd(M1, GO)) : sadd(hw::pk_min_u(sadd(I1, GE), sadd(M1, GO)), NUC);
                    I2 = shift_down<B>(INFB, ish, li);                                      // :318-319
                    if constexpr (TR) {
                        const uint32_t tm = M2 & 0x00030003u, ti = I2 & 0x00030003u, td = D2 & 0x00030003u;
                        M2 ^= tm; I2 |= 0x00010001u; D2 |= 0x00030003u;
                        const uint32_t bpo = hw_lshl_or(td, 4, hw_lshl_or(ti, 2, tm));
                        if (OCT_DP_ABLATE != 5) tile[(k & 15) * kTileStride + lane] = hw_lshl_or(bpo, 6, bpe | flag_of(fo, 14));
#if !defined(OCTPHMM_SIM)
                        else asm volatile("" :: "v"(bpo | bpe));
#endif
                    }
                    if constexpr (PAIRED) { rr = rr_nx; cA.x = nA.x; nA.x = e1; GOn = e4.x; GEn = e4.y; GO = e4.z; GE = e4.w; }
                    else {
                    rr = rr_nx; cA = nA; cB = nB; GO = GOn; GE = GEn; nA = nnA; nB = nnB;
                    GOn = hw::perm(nB.y, nA.y, 0x05040100u); GEn = hw::perm(nB.y, nA.y, 0x07060302u);
                    }
                }
                if constexpr (TR) {
                    if (((k0 + 4) & 15) == 0) flush_tile(k0 >> 4);
                }
            };

            const uint32_t kB = (Tmin & ~3u) > (uint32_t)B ? (Tmin & ~3u) : (uint32_t)B;   // no end cell before the shortest read is consumed
            uint32_t k = 0;
            // quads up to iteration k_end; with p.rec_chunk the row is restaged (and rp rebased) whenever k reaches the end of what it holds
            uint32_t k_restage = p.rec_chunk ? p.rec_chunk : 0xffffffffu;
            auto run_to = [&](const uint32_t k_end, auto init_c, auto cap_c, auto tr_c) {
                while (k < k_end) {
                    if (k == k_restage) { stage(k); rp = rec_row + (B - li) - k; k_restage += p.rec_chunk; }
                    const uint32_t stop = k_end < k_restage ? k_end : k_restage;
                    for (; k < stop; k += 4) quad(k, init_c, cap_c, tr_c);
                }
            };
            if constexpr (TRACE) {
                // Late traceback start (p.late): every task of this launch needs its walk only inside the RIGHT inactive flank, i.e. only the
                // traceback words of the last iterations. Until k_sw the wave runs the score-only recurrence (no label extraction, no tile
                // writes); at k_sw the states get the labels they would carry (M 0, I 1, D 3, :63-65) and the traceback form takes over.
                // Scores are the same in both forms (labels only break ties inside an iteration and are reset after it); k_sw is a tile
                // boundary no later than the first possible end cell, so end-cell ties see their labels.
                uint32_t k_sw = 0;
                if (p.late && g * G >= late_from) {
                    // (a task that also touches the LEFT flank - its window starts before the flank's end, off < reg_lhs: k_classify's class 2 - needs its whole
                    // traceback: its first needed iteration is 0, and so is the group's)
                    auto first_needed = [&](const DevTask& t, uint32_t T) -> uint32_t {
                        const uint32_t reg = p.hap_region[t.hap];
                        if (t.off < p.reg_lhs[reg]) return 0u;
                        const uint32_t Lh = p.hoff[t.hap + 1] - p.hoff[t.hap], L = T + 2 * B - 1, rhs = p.reg_rhs[reg];
                        const uint32_t rhs_w = t.off + L + rhs < Lh ? 0u : (t.off + L + rhs - Lh < L ? t.off + L + rhs - Lh : L);   // right flank in window coordinates (pair_hmm.hpp:580-587)
                        const uint32_t rhs_begin = L - rhs_w;
                        return rhs_begin > (uint32_t)B + 2 ? rhs_begin - B - 2 : 0u;        // cells with x >= rhs_begin - 2 lie on iterations k >= x - (B - 1) - 1
                    };
                    uint32_t ks = first_needed(tA, TA); const uint32_t kb2 = first_needed(tB, TB);
                    ks = kb2 < ks ? kb2 : ks;
                    for (int m = B; m < 64; m <<= 1) { const uint32_t o = hw::shfl_xor(ks, m); ks = o < ks ? o : ks; }
                    ks = hw::readfirstlane(ks);
                    k_sw = (ks < Tmin ? ks : Tmin) & ~15u;
                }
                run_to(k_sw < (uint32_t)B ? k_sw : (uint32_t)B, BoolC<true>{}, BoolC<false>{}, BoolC<false>{});
                run_to(k_sw, BoolC<false>{}, BoolC<false>{}, BoolC<false>{});
                if (k_sw) { I1 |= 0x00010001u; D1 |= 0x00030003u; I2 |= 0x00010001u; D2 |= 0x00030003u; y1 = hw::pk_min_u(M1, I1); }
            }
            if (Tmin >= (uint32_t)B) run_to((uint32_t)B, BoolC<true>{}, BoolC<false>{}, BoolC<true>{});   // no read of the wave ends inside the rolling initialisation: no end cells to capture yet
            run_to((uint32_t)B, BoolC<true>{}, BoolC<true>{}, BoolC<true>{});                              // rolling initialisation lasts B iterations
            run_to(kB, BoolC<false>{}, BoolC<false>{}, BoolC<true>{});
            run_to(K4, BoolC<false>{}, BoolC<true>{}, BoolC<true>{});
            if constexpr (TRACE) { if (K4 & 15) flush_tile(K4 >> 4); }

            // ---- first minimum over the row's end cells, per packed task (:285-291,309-315,323) ----
            if (OCT_DP_ABLATE == 3 && TRACE && li == 0) { TraceEnd e; e.score = 0; e.sidx = -1; p.ends[g * G + 2 * row] = e; p.ends[g * G + 2 * row + 1] = e; }   // (no walk)
            for (uint32_t half = 0; half < (OCT_DP_ABLATE == 3 ? 0u : 2u); ++half) {
                const uint32_t Th = half ? TB : TA;
                const uint32_t vE = (bestE >> (16 * half)) & 0xffffu, vO = (bestO >> (16 * half)) & 0xffffu;   // biased: unsigned order
                const uint32_t sE = 2 * (Th + li);
                const uint32_t kE = vE << 16 | sE, kO = vO << 16 | (sE + 1);
                uint32_t key = kE < kO ? kE : kO;
                for (int m = 1; m < B; m <<= 1) { const uint32_t o = hw::shfl_xor(key, m); key = o < key ? o : key; }
                if (li == 0) {
                    const DevTask& t = half ? tB : tA;
                    const int32_t score = (int32_t)((key >> 16) >> 2);                      // (minscore - null_score_) >> 2
                    if constexpr (TRACE) {
                        TraceEnd e; e.score = score; e.sidx = (key >> 16) >= (0x7800u ^ 0x8000u) ? -1 : (int32_t)(key & 0xffffu);
                        p.ends[g * G + 2 * row + half] = e;
                    } else {
                        if (t.pair != kPadTask) hw::atomic_min_i32(p.pair_best + t.pair, score);
                    }
                }
            }
            hw::wave_lds_fence();
        };
        for (uint32_t g = seg + wave; g < seg_end; g += kBlockWaves) {
            if constexpr (CAN_PAIR) { if (paired) run_group(BoolC<true>{}, g); else run_group(BoolC<false>{}, g); }
            else run_group(BoolC<false>{}, g);
        }
        seg = seg_end;
    }
    }
}

// A traceback launch heads a chain (DP -> walk -> epilogue) while the score-only launch beside it has nobody waiting: where their waves share a SIMD the
// traceback wave issues first (s_setprio; OCT_CHAIN_PRIO=0 builds without, for A/B).
#ifndef OCT_CHAIN_PRIO
#define OCT_CHAIN_PRIO 2
#endif
template <bool HEAD> OCT_DEVICE void chain_head_priority() { if constexpr (HEAD && OCT_CHAIN_PRIO > 0) hw::wave_priority<OCT_CHAIN_PRIO>(); }

template <int B, bool TRACE, bool GENERIC, bool FASTADD>
OCT_KERNEL(k_dp)(DpParams p) { chain_head_priority<TRACE>(); dp_groups<B, TRACE, GENERIC, FASTADD>(p, hw::block_idx(), hw::grid_dim()); }

// Region-sized (device-sized) steps: the traceback list and the score-only list of one cost flavour in ONE launch - the first n_blocks_t workgroups take the
// traceback form, the rest the score-only form. Both are a few hundred latency-bound waves: side by side they fill the chip's SIMDs once, one after the other
// (or on two streams, with an event between them) they cost a launch gap and their sum.
template <int B, bool GENERIC, bool FASTADD>
OCT_KERNEL(k_dp_pair)(DpParams pt, DpParams ps, uint32_t n_blocks_t)
{
    if (hw::block_idx() < n_blocks_t) { chain_head_priority<true>(); dp_groups<B, true, GENERIC, FASTADD>(pt, hw::block_idx(), n_blocks_t); }
    else dp_groups<B, false, GENERIC, FASTADD>(ps, hw::block_idx() - n_blocks_t, hw::grid_dim() - n_blocks_t);
}

// ------------------------------------------------------------------------------------------------------------------
// int32 lanes (Config::use_int_scores, "--use-wide-hmm-scores"): one task per band row, plain 32-bit VALU, generic byte tests.
// Same staging, tiles and traceback word format as k_dp (labels in the low 16 bits), so k_walk<B, 1> serves it unchanged.
// Deliberately simple (no loop phases): this is the reference's slow precision mode, not the throughput path.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t INF32 = 0x7ffff800u;      // INT_MAX - 0x7FF (simd_pair_hmm.hpp:55-56 with ScoreType = int)
constexpr uint32_t NUL32 = 0x80000000u;      // INT_MIN

OCT_DEVICE uint32_t min_i32(uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b ? a : b; }

template <int B> OCT_DEVICE uint32_t shift_up32(uint32_t v, uint32_t li) { return shift_up<B>(INF32, v, li); }
template <int B> OCT_DEVICE uint32_t shift_down32(uint32_t v, uint32_t li) { return shift_down<B>(INF32, v, li); }

template <int B, bool TRACE>
OCT_KERNEL(k_dp32)(DpParams p)
{
    constexpr uint32_t ROWS = 64 / B, G = ROWS;
    OCT_DYN_SMEM(smem);
    const uint32_t tid = hw::thread_idx(), lane = tid & 63, wave = tid >> 6;
    const uint32_t row = lane / B, li = lane % B;
    const uint32_t lh_n = (p.lh_cap + 8 + 1) & ~1u, rec_n = dp_rec_n(p.t_cap, B);
    uint2* tabF = (uint2*)smem;
    uint2* tabR = tabF + lh_n;
    uint2* recs = tabR + lh_n;
    uint32_t* tiles = (uint32_t*)(recs + kBlockWaves * ROWS * rec_n);
    uint2* rec_row = recs + (wave * ROWS + row) * rec_n;
    uint32_t* tile = tiles + wave * 16 * kTileStride;
    const DevTask* tasks = p.tasks; uint32_t n_tasks = p.n_tasks;
    if (p.ref.totals) { uint32_t first; task_list_range(p.ref, first, n_tasks); tasks += first; }   // device-sized launch (see k_dp)
    const uint32_t n_groups = n_tasks / G;
    const uint32_t NUC = p.nuc4 & 0xffffu; const uint32_t NUC32 = (uint32_t)(int32_t)(int16_t)NUC;   // sign-extended (nuc_prior << 2)

    for (uint32_t g_begin = hw::block_idx() * p.groups_per_block; g_begin < n_groups; g_begin += hw::grid_dim() * p.groups_per_block) {
    const uint32_t g_end = g_begin + p.groups_per_block < n_groups ? g_begin + p.groups_per_block : n_groups;
    uint32_t seg = g_begin;
    while (seg < g_end) {
        const uint32_t hap = tasks[seg * G].hap;
        uint32_t seg_end = seg + 1;
        while (seg_end < g_end && tasks[seg_end * G].hap == hap) ++seg_end;
        const uint32_t ho = p.hoff[hap], Lh = p.hoff[hap + 1] - ho;
        hw::block_sync();
        for (uint32_t x = tid; x < lh_n; x += kBlockWaves * 64) {
            const bool in = x < Lh;
            tabF[x] = in ? p.tabF[ho + x] : make_uint2(0, 0);
            tabR[x] = in ? p.tabR[ho + x] : make_uint2(0, 0);
        }
        hw::block_sync();
        for (uint32_t g = seg + wave; g < seg_end; g += kBlockWaves) {
            const DevTask tA = tasks[g * G + row];
            const uint32_t roA = p.roff[tA.read], TA = p.roff[tA.read + 1] - roA;
            uint32_t Tmax = TA;
            for (int m = B; m < 64; m <<= 1) { const uint32_t o = hw::shfl_xor(Tmax, m); Tmax = o > Tmax ? o : Tmax; }
            Tmax = hw::readfirstlane(Tmax);
            const uint32_t K = Tmax + B, K16 = (K + 15) & ~15u;
            for (uint32_t j = li; j < K + B + 2; j += B) {            // read-side records: {target char (0x100 / '0' padding), quality << 2}
                const int32_t t = (int32_t)j - B;
                const bool in = t >= 0 && (uint32_t)t < TA;
                const uint32_t r = in ? ld8(p.rbases + roA + t) : (t < 0 ? 0x100u : (uint32_t)'0');
                const uint32_t q = in ? ld8(p.rquals + roA + t) : 64u;
                rec_row[j] = make_uint2(r, q << 2);
            }
            hw::wave_lds_fence();
            const uint2* pA = (p.rrev[tA.read] ? tabR : tabF) + tA.off + li;
            const uint2* rp = rec_row + (B - li);
            const uint32_t kend = TA + li;
            uint4* bpg = TRACE ? (uint4*)(p.bp + (size_t)g * p.k_cap * 1024) : nullptr;
            uint32_t M1 = INF32, I1 = INF32, D1 = INF32, M2 = INF32, I2 = INF32, D2 = INF32, bestE = INF32, bestO = INF32;
            auto cost = [&](const uint2 r2, const uint2 a, uint32_t* mism) -> uint32_t {      // update_match_state :121-132 on raw bytes
                const uint32_t h = a.x & 0xffu, m = (a.x >> 8) & 0xffu, p4 = ((a.x >> 16) & 0xffu) << 2, isn = a.x >> 24;
                const uint32_t inner = r2.x == m ? p4 : r2.y;
                uint32_t c = min_i32(r2.y, inner);
                if (r2.x == h) c = 0;
                *mism = r2.x != h ? 1u : 0u;
                return min_i32(c, isn ? 8u : INF32);
            };
            auto flush_tile = [&](uint32_t kt) {
                hw::wave_lds_fence();
                uint4* dst = bpg + (size_t)kt * 256;
                for (int j = 0; j < 4; ++j) {
                    const uint32_t src = 16 * j + (lane >> 2), kq = (lane & 3) * 4;
                    dst[j * 64 + lane] = make_uint4(tile[(kq + 0) * kTileStride + src], tile[(kq + 1) * kTileStride + src],
                                                    tile[(kq + 2) * kTileStride + src], tile[(kq + 3) * kTileStride + src]);
                }
                hw::wave_lds_fence();
            };
            for (uint32_t k = 0; k < K16; ++k) {
                if (k < K) {
                    const uint2 rr = rp[k], cA = pA[k], nA = pA[k + 1];
                    const uint32_t GO = cA.y & 0xffffu, GE = cA.y >> 16, GOn = nA.y & 0xffffu, GEn = nA.y >> 16;
                    if (k == li) { M1 = NUL32; M2 = NUL32; }                                // rolling initialiser (k == li < B)
                    uint32_t mismE, mismO;
                    const uint32_t m1 = min_i32(M1, min_i32(I1, D1));                       // :284
                    if (k == kend) bestE = min_i32(bestE, m1);                              // :285-291
                    M1 = m1 + cost(rr, cA, &mismE);                                         // :292
                    const uint32_t x2 = min_i32(M2, I2);
                    D1 = shift_up32<B>(min_i32(D2 + GEn, x2 + GOn), li);                    // :293-294
                    I1 = min_i32(I2 + GE, M2 + GO) + NUC32;                                 // :295
                    uint32_t bpe = 0;
                    if constexpr (TRACE) {
                        const uint32_t tm = M1 & 3u, ti = I1 & 3u, td = D1 & 3u;
                        M1 ^= tm; I1 = (I1 & ~3u) | 1u; D1 |= 3u;
                        bpe = tm | ti << 2 | td << 4 | mismE << 15;
                    }
                    const uint32_t m2 = min_i32(x2, D2);                                    // :308
                    if (k == kend) bestO = min_i32(bestO, m2);
                    M2 = m2 + cost(rr, nA, &mismO);                                         // :316
                    const uint32_t y1 = min_i32(M1, I1);
                    D2 = min_i32(D1 + GEn, y1 + GOn);                                       // :317
                    I2 = shift_down32<B>(min_i32(I1 + GE, M1 + GO) + NUC32, li);            // :318-319
                    if constexpr (TRACE) {
                        const uint32_t tm = M2 & 3u, ti = I2 & 3u, td = D2 & 3u;
                        M2 ^= tm; I2 = (I2 & ~3u) | 1u; D2 |= 3u;
                        tile[(k & 15) * kTileStride + lane] = bpe | (tm | ti << 2 | td << 4) << 6 | mismO << 14;
                    }
                }
                if constexpr (TRACE) { if ((k & 15) == 15) flush_tile(k >> 4); }
            }
            // first minimum over the row's end cells (:285-291,309-315,323): 64-bit key = (biased value, diagonal index)
            const uint32_t vE = bestE ^ 0x80000000u, vO = bestO ^ 0x80000000u, sE = 2 * (TA + li);
            uint32_t kv = vE, ks = sE;
            if (vO < vE) { kv = vO; ks = sE + 1; }
            for (int m = 1; m < B; m <<= 1) {
                const uint32_t ov = hw::shfl_xor(kv, m), os = hw::shfl_xor(ks, m);
                if (ov < kv || (ov == kv && os < ks)) { kv = ov; ks = os; }
            }
            if (li == 0) {
                const int32_t score = (int32_t)kv >> 2;                                     // (minscore - INT_MIN) >> 2 in wrapping int arithmetic
                if constexpr (TRACE) {
                    TraceEnd e; e.score = score; e.sidx = kv >= (INF32 ^ 0x80000000u) ? -1 : (int32_t)ks;
                    p.ends[g * G + row] = e;
                } else {
                    if (tA.pair != kPadTask) hw::atomic_min_i32(p.pair_best + tA.pair, score);
                }
            }
            hw::wave_lds_fence();
        }
        seg = seg_end;
    }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Wide / long path: band 64 x C (C = 1, 2, 4 adjacent diagonals per lane, one task per wave), int16 or int32 semantics, operands streamed
// from HBM/L2 instead of staged in LDS. Serves bands 128 and 256 (simd_pair_hmm_wrapper.hpp:207-208) and reads/haplotypes too long
// for the LDS-resident kernels (BASELINE.json configs[4]: 10 kb reads, 20 kb haplotypes, band 256, int32 lanes). Plain 32-bit VALU with the
// reference's wrap emulated by a sign-extension after every add when W16; generic byte tests; same traceback word format, written
// as [tile][plane c][lane][16 iterations] so that k_walk<B, 1, C> fetches one 64-byte line per 16 steps of a diagonal.
// ------------------------------------------------------------------------------------------------------------------
template <bool W16> OCT_DEVICE uint32_t wadd(uint32_t a, uint32_t b)
{
    if constexpr (W16) return (uint32_t)(int32_t)(int16_t)(uint16_t)(a + b); else return a + b;
}

template <int B, bool TRACE, bool W16>
OCT_KERNEL(k_dp_wide)(DpParams p)
{
    chain_head_priority<TRACE>();
    // B >= 64: one task per wave, C = B / 64 adjacent diagonals per lane. B < 64 (long reads at a narrow band - the reference's PacBio
    // configuration runs band 16): 64 / B tasks per wave, one per row of B lanes, exactly the task-group layout of the LDS-resident kernels.
    constexpr int C = B > 64 ? B / 64 : 1, BL = B < 64 ? B : 64, ROWS = 64 / BL;
    constexpr uint32_t INFW = W16 ? 0x00007800u : INF32;
    constexpr uint32_t NULW = W16 ? 0xffff8000u : NUL32;
    const uint32_t tid = hw::thread_idx(), lane = tid & 63, wave = tid >> 6;
    const uint32_t row = lane / BL, li = lane % BL;
    const uint32_t group = hw::block_idx() * kBlockWaves + wave;           // = the wave's task group (ROWS tasks)
    const DevTask* tasks = p.tasks; uint32_t n_tasks = p.n_tasks;
    if (p.ref.totals) { uint32_t first; task_list_range(p.ref, first, n_tasks); tasks += first; }   // device-sized launch: the grid is the host's bound
    if (group * ROWS >= n_tasks) return;                                   // whole waves only (n_tasks is a multiple of ROWS)
    const uint32_t task = group * ROWS + row;
    const DevTask t = tasks[task];
    const uint32_t ro = p.roff[t.read], T = p.roff[t.read + 1] - ro;
    uint32_t K = T + B;
    if (ROWS > 1) { for (int m = BL; m < 64; m <<= 1) { const uint32_t o = hw::shfl_xor(K, m); K = o > K ? o : K; } K = hw::readfirstlane(K); }   // the rows iterate together
    const uint32_t last_rec = T + 2 * B;                                   // operand index one past this task's window (already read by the B >= 64 form)
    const uint2* tab = (p.rrev[t.read] ? p.tabR : p.tabF) + p.hoff[t.hap] + t.off;   // generic table of the band window
    const uint32_t i0 = li * C;                                            // this lane's first band diagonal
    const uint32_t NUCW = (uint32_t)(int32_t)(int16_t)(p.nuc4 & 0xffffu);
    uint32_t M1[C], I1[C], D1[C], M2[C], I2[C], D2[C];
    for (int c = 0; c < C; ++c) M1[c] = I1[c] = D1[c] = M2[c] = I2[c] = D2[c] = INFW;
    uint32_t best = INFW, best_s = 0; bool have = false;
    // operand windows: hap records x = k + i0 + c (c = 0..C), read records t = k - i0 - c (c = 0..C-1)
    uint2 hw_[C + 1];
    for (int c = 0; c <= C; ++c) hw_[c] = tab[i0 + c];
    auto read_rec = [&](int32_t tt) -> uint2 {
        if (tt < 0) return make_uint2(0x100u, 64u << 2);
        if ((uint32_t)tt >= T) return make_uint2((uint32_t)'0', 64u << 2);
        return make_uint2(ld8(p.rbases + ro + tt), ld8(p.rquals + ro + tt) << 2);
    };
    uint2 rw[C];
    for (int c = 0; c < C; ++c) rw[c] = read_rec(-(int32_t)i0 - c);
    // Operand prefetch: a lone wave per SIMD cannot hide a memory round trip per iteration, so the haplotype records and read bytes of the
    // NEXT 8 iterations are fetched while the current 8 compute (registers: 8 table entries + 8 read bases + 8 qualities per lane).
    constexpr int CH = 8;
    struct Chunk { uint2 tab[CH]; uint64_t rb, rq; int32_t tt0, start; };
    auto fetch = [&](uint32_t k0) -> Chunk {                             // operands consumed at the END of iterations k0 .. k0 + 7
        Chunk ch;
        for (int e = 0; e < CH; ++e) { const uint32_t nx = k0 + e + 1 + i0 + C; ch.tab[e] = tab[nx > last_rec ? last_rec : nx]; }
        ch.tt0 = (int32_t)(k0 + 1) - (int32_t)i0;                         // read position of element 0
        const int32_t hi = (int32_t)T > CH ? (int32_t)T - CH : 0;
        ch.start = ch.tt0 < 0 ? 0 : (ch.tt0 > hi ? hi : ch.tt0);          // 8 bytes from a position inside the read (the batch's arrays carry a 16-byte tail pad)
        ch.rb = ld64u(p.rbases + ro + ch.start); ch.rq = ld64u(p.rquals + ro + ch.start);
        return ch;
    };
    auto chunk_read_rec = [&](const Chunk& ch, int e) -> uint2 {
        const int32_t tt = ch.tt0 + e;
        if (tt < 0) return make_uint2(0x100u, 64u << 2);
        if ((uint32_t)tt >= T) return make_uint2((uint32_t)'0', 64u << 2);
        const uint32_t sh = 8u * (uint32_t)(tt - ch.start);
        return make_uint2((uint32_t)(ch.rb >> sh) & 0xffu, ((uint32_t)(ch.rq >> sh) & 0xffu) << 2);
    };
    auto cost = [&](const uint2 r2, const uint2 a, uint32_t* mism) -> uint32_t {           // update_match_state :121-132
        const uint32_t h = a.x & 0xffu, m = (a.x >> 8) & 0xffu, p4 = ((a.x >> 16) & 0xffu) << 2, isn = a.x >> 24;
        const uint32_t inner = r2.x == m ? p4 : r2.y;
        uint32_t c = min_i32(r2.y, inner);
        if (r2.x == h) c = 0;
        *mism = r2.x != h ? 1u : 0u;
        return min_i32(c, isn ? 8u : INFW);
    };
    uint32_t* bpt = TRACE ? p.bp + (size_t)group * p.k_cap * C * 1024 : nullptr;
    Chunk nxt = fetch(0);
    for (uint32_t k0 = 0; k0 < K; k0 += CH) {
      const Chunk cur = nxt;
      if (k0 + CH < K) nxt = fetch(k0 + CH);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const uint32_t k = k0 + (uint32_t)e;                              // iterations past K (K is not a multiple of 8) only touch spare traceback words
        uint32_t dsh[C], ish[C], bpe[C];
        for (int c = 0; c < C; ++c) {
            const uint32_t i = i0 + c;
            const uint2 cA = hw_[c], nA = hw_[c + 1];
            const uint32_t GO = cA.y & 0xffffu, GE = cA.y >> 16, GOn = nA.y & 0xffffu, GEn = nA.y >> 16;
            if (k == i) { M1[c] = NULW; M2[c] = NULW; }                                     // rolling initialiser
            uint32_t mE, mO;
            const uint32_t m1 = min_i32(M1[c], min_i32(I1[c], D1[c]));                      // :284
            if (k == T + i && (int32_t)m1 < (int32_t)best) { best = m1; best_s = 2 * k; have = true; }   // :285-291 (k - T == i)
            M1[c] = wadd<W16>(m1, cost(rw[c], cA, &mE));                                    // :292
            const uint32_t x2 = min_i32(M2[c], I2[c]);
            dsh[c] = min_i32(wadd<W16>(D2[c], GEn), wadd<W16>(x2, GOn));                    // :293 (shifted below)
            I1[c] = wadd<W16>(min_i32(wadd<W16>(I2[c], GE), wadd<W16>(M2[c], GO)), NUCW);   // :295
            bpe[c] = mE << 15;
            (void)mO;
        }
        {   // :294 D1 <- shifted one diagonal up, infinity_ into diagonal 0 (of every row)
            uint32_t from_below = hw::dpp_wave_shr1(INFW, dsh[C - 1]);
            if (ROWS > 1 && li == 0) from_below = INFW;
            for (int c = C - 1; c >= 1; --c) D1[c] = dsh[c - 1];
            D1[0] = from_below;
        }
        if constexpr (TRACE) {
            for (int c = 0; c < C; ++c) {
                const uint32_t tm = M1[c] & 3u, ti = I1[c] & 3u, td = D1[c] & 3u;
                M1[c] ^= tm; I1[c] = (I1[c] & ~3u) | 1u; D1[c] |= 3u;
                bpe[c] |= tm | ti << 2 | td << 4;
            }
        }
        for (int c = 0; c < C; ++c) {
            const uint32_t i = i0 + c;
            const uint2 cA = hw_[c], nA = hw_[c + 1];
            const uint32_t GO = cA.y & 0xffffu, GE = cA.y >> 16, GOn = nA.y & 0xffffu, GEn = nA.y >> 16;
            uint32_t mO;
            const uint32_t m2 = min_i32(min_i32(M2[c], I2[c]), D2[c]);                      // :308
            if (k == T + i && (int32_t)m2 < (int32_t)best) { best = m2; best_s = 2 * k + 1; have = true; }
            M2[c] = wadd<W16>(m2, cost(rw[c], nA, &mO));                                    // :316
            const uint32_t y1 = min_i32(M1[c], I1[c]);
            D2[c] = min_i32(wadd<W16>(D1[c], GEn), wadd<W16>(y1, GOn));                     // :317
            ish[c] = wadd<W16>(min_i32(wadd<W16>(I1[c], GE), wadd<W16>(M1[c], GO)), NUCW);  // :318 (shifted below)
            bpe[c] |= mO << 14;
        }
        {   // :318-319 I2 <- shifted one diagonal down, infinity_ into the last diagonal (of every row)
            uint32_t from_above = hw::dpp_wave_shl1(INFW, ish[0]);
            if (ROWS > 1 && li == (uint32_t)BL - 1) from_above = INFW;
            for (int c = 0; c < C - 1; ++c) I2[c] = ish[c + 1];
            I2[C - 1] = from_above;
        }
        if constexpr (TRACE) {
            for (int c = 0; c < C; ++c) {
                const uint32_t tm = M2[c] & 3u, ti = I2[c] & 3u, td = D2[c] & 3u;
                M2[c] ^= tm; I2[c] = (I2[c] & ~3u) | 1u; D2[c] |= 3u;
                bpt[(((size_t)(k >> 4) * C + c) * 64 + lane) * 16 + (k & 15)] = bpe[c] | (tm | ti << 2 | td << 4) << 6;
            }
        }
        // slide the operand windows by one position (past its own window a task re-reads its last record)
        for (int c = 0; c < C; ++c) hw_[c] = hw_[c + 1];
        hw_[C] = cur.tab[e];
        for (int c = C - 1; c >= 1; --c) rw[c] = rw[c - 1];
        rw[0] = chunk_read_rec(cur, e);
      }
    }
    // first minimum over the end cells: per lane the candidates were visited in increasing diagonal order, so strict < kept the first
    uint32_t kv = have ? (W16 ? ((best + 0x8000u) & 0xffffu) : (best ^ 0x80000000u)) : 0xffffffffu, ks = best_s;
    if (!have) ks = 0xffffffffu;
    for (int m = 1; m < BL; m <<= 1) {
        const uint32_t ov = hw::shfl_xor(kv, m), os = hw::shfl_xor(ks, m);
        if (ov < kv || (ov == kv && os < ks)) { kv = ov; ks = os; }
    }
    if (li == 0) {
        // no end cell below infinity_: minscore stays infinity_, minscoreidx -1 (:269-270)
        const bool none = kv == 0xffffffffu;
        const uint32_t biased = none ? (W16 ? ((INFW + 0x8000u) & 0xffffu) : (INFW ^ 0x80000000u)) : kv;
        const int32_t score = W16 ? (int32_t)(biased >> 2) : ((int32_t)biased >> 2);
        if constexpr (TRACE) { TraceEnd e; e.score = score; e.sidx = none ? -1 : (int32_t)ks; p.ends[task] = e; }
        else if (t.pair != kPadTask) hw::atomic_min_i32(p.pair_best + t.pair, score);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Long reads at band 16 with int32 lanes (unsplit PacBio reads at max-indel-errors 16: the realignment path): four tasks per wave, one per DPP row of 16
// lanes, operands streamed - k_dp_wide<16, *, false> rebuilt (round 4, step 8). k_dp_wide made the read operand of every iteration out of two 64-bit words with per-lane shifts
// and range tests (12 instructions), took the generic byte-test cost for every task (2 x 12), tested for the rolling initialiser and the end cells in every iteration, and every
// lane loaded its own table entry per iteration (each entry of a window 16 times over): 82 vector instructions and 768 B of loads per wave-iteration. Here
//   * the read operand is one word of the read's record row (DevBatch::rrecW: v_perm selector | padded base << 8 | quality << 24), pure-ACGT reads on clean haplotypes take the
//     fast cost (perm + min; the lists of k_classify as on every other path), and the loop runs in three phases (initialiser / steady / end cells);
//   * the operands TRAVEL along the row instead of being loaded per lane: lane i's table entry of iteration k + 1 is lane i + 1's of iteration k, its read record lane i - 1's -
//     one DPP row shift each, the row's last / first lane taking the one new value. The new values of sixteen iterations are ONE coalesced load per lane (entry k0 + 17 + lane,
//     record k0 + 1 + lane), rotated down the row by one lane per iteration so that the taker finds its value at the row's first lane: a sixteenth of the loads, and no chunk of
//     eight entries per lane in registers (the first form of this kernel kept two: 124 registers and spills).
// Same recurrence, same traceback words and tile layout as k_dp_wide<16, TRACE, false>: k_walk<16, 1, 1> and k_walk_rows<16, 1> serve both.
// ------------------------------------------------------------------------------------------------------------------
template <bool TRACE, bool GENERIC>
OCT_KERNEL(k_dp_rows)(DpParams p)
{
    chain_head_priority<TRACE>();
    constexpr int B = 16, ROWS = 4, CH = 8;
    const uint32_t tid = hw::thread_idx(), lane = tid & 63, wave = tid >> 6;
    const uint32_t row = lane / B, li = lane % B;
    const DevTask* tasks = p.tasks; uint32_t n_tasks = p.n_tasks;
    if (p.ref.totals) { uint32_t first; task_list_range(p.ref, first, n_tasks); tasks += first; }   // device-sized launch: the grid is the host's bound
    const uint32_t group = hw::block_idx() * (hw::block_dim() >> 6) + wave;   // = the wave's task group (ROWS tasks); the launcher picks the workgroup's waves
    if (group * ROWS >= n_tasks) return;                                   // whole waves only (n_tasks is a multiple of ROWS)
    const uint32_t task = group * ROWS + row;
    const DevTask t = tasks[task];
    const uint32_t ro = p.roff[t.read], T = p.roff[t.read + 1] - ro;
    uint32_t t_hi = T, t_lo = T;                                           // the rows iterate together: to the longest read's end, the end-cell phase from the shortest's
    for (int m = B; m < 64; m <<= 1) { const uint32_t a = hw::shfl_xor(t_hi, m), c = hw::shfl_xor(t_lo, m); t_hi = a > t_hi ? a : t_hi; t_lo = c < t_lo ? c : t_lo; }
    t_hi = hw::readfirstlane(t_hi); t_lo = hw::readfirstlane(t_lo);
    const uint32_t KC = (t_hi + (uint32_t)B + 15u) & ~15u;                 // whole tiles of 16 iterations
    const uint2* tab = (p.rrev[t.read] ? p.tabR : p.tabF) + p.hoff[t.hap] + t.off;
    const uint32_t* rrow = p.rrecW + (size_t)t.read * p.rrec_stride;       // entry j = read position j - B: selector | base << 8 | quality << 24
    const uint32_t NUCW = (uint32_t)(int32_t)(int16_t)(p.nuc4 & 0xffffu);
    uint32_t M1 = INF32, I1 = INF32, D1 = INF32, M2 = INF32, I2 = INF32, D2 = INF32;
    uint32_t best = INF32, best_s = 0; bool have = false;
    // operand indices are clamped to the task's own window (+ 1) and to the record row: a row whose own iterations are over (a shorter read than the wave's longest) keeps
    // re-reading its last entry - those iterations feed no end cell
    const uint32_t e_max = T + 2 * (uint32_t)B, r_max = p.rrec_stride - 1;
    auto tab_at = [&](uint32_t e) -> uint2 { return tab[e < e_max ? e : e_max]; };
    auto rec_at = [&](uint32_t j) -> uint32_t { return rrow[j < r_max ? j : r_max]; };
    uint2 cE = tab_at(li), nE = tab_at(li + 1);                            // table entries of x = k + li and x + 1
    uint32_t rw = rec_at((uint32_t)B - li);                                // read record of t = k - li
    uint2 Fe = tab_at(17u + li); uint32_t Fr = rec_at((uint32_t)B + 1u + li);   // the sixteen values this tile's iterations hand to the rows' last / first lanes, element u at lane u
    uint32_t GO = cE.y & 0xffffu, GE = cE.y >> 16;
    auto cost = [&](const uint32_t rec, const uint2 a, uint32_t& flag) -> uint32_t {   // update_match_state (:121-132), already shifted by the trace bits
        if constexpr (GENERIC) {
            const uint32_t rc = (rec >> 8) & 0x1ffu, q4 = (rec >> 24) << 2;
            const uint32_t h = a.x & 0xffu, m = (a.x >> 8) & 0xffu, p4 = ((a.x >> 16) & 0xffu) << 2, isn = a.x >> 24;
            const uint32_t inner = rc == m ? p4 : q4;
            uint32_t c = min_i32(q4, inner);
            if (rc == h) c = 0;
            flag = rc != h ? 0x8000u : 0u;
            return min_i32(c, isn ? 8u : INF32);
        } else {
            const uint32_t cp = hw::perm(a.x, a.x, rec) & 0xffu;                   // this read base's cap at this haplotype position (0xff: none); only selector byte 0 counts
            const uint32_t q = rec >> 24;
            const uint32_t c = cp < q ? cp : q;                                    // min(quality, cap)
            flag = c ? 0x8000u : 0u;                                               // the walk charges exactly c inside a flank
            return c << 2;
        }
    };
    uint32_t* const bpt = TRACE ? p.bp + (size_t)group * p.k_cap * 1024 + (size_t)lane * 16 : nullptr;   // this lane's 16-word line in tile 0 (+ tile * 1024 + (k & 15))
    auto run_chunk = [&](uint32_t k0, auto init_c, auto cap_c) __attribute__((always_inline)) {
        constexpr bool INIT = decltype(init_c)::value, CAP = decltype(cap_c)::value;
        uint32_t bw[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const uint32_t k = k0 + (uint32_t)u;
            const uint32_t GOn = nE.y & 0xffffu, GEn = nE.y >> 16;
            if constexpr (INIT) { if (k == li) { M1 = NUL32; M2 = NUL32; } }                  // rolling initialiser
            const uint32_t x2 = min_i32(M2, I2);
            const uint32_t dsh = min_i32(D2 + GEn, x2 + GOn);                                 // :293
            uint32_t fE, fO;
            const uint32_t m1 = min_i32(M1, min_i32(I1, D1));                                 // :284
            if constexpr (CAP) { if (k == T + li && (int32_t)m1 < (int32_t)best) { best = m1; best_s = 2 * k; have = true; } }   // :285-291
            M1 = m1 + cost(rw, cE, fE);                                                       // :292
            I1 = min_i32(I2 + GE, M2 + GO) + NUCW;                                            // :295
            uint32_t bpe = 0;
            if constexpr (TRACE) { const uint32_t tm = M1 & 3u, ti = I1 & 3u; M1 ^= tm; I1 = (I1 & ~3u) | 1u; bpe = tm | ti << 2 | fE; }
            const uint32_t ish = min_i32(I1 + GE, M1 + GO) + NUCW;                            // :318
            D1 = hw::dpp_row_shr1(INF32, dsh);                                                // :294 one diagonal up inside the row, infinity_ into its first diagonal
            if constexpr (TRACE) { const uint32_t td = D1 & 3u; D1 |= 3u; bpe |= td << 4; }
            const uint32_t m2 = min_i32(x2, D2);                                              // :308
            if constexpr (CAP) { if (k == T + li && (int32_t)m2 < (int32_t)best) { best = m2; best_s = 2 * k + 1; have = true; } }
            M2 = m2 + cost(rw, nE, fO);                                                       // :316
            const uint32_t y1 = min_i32(M1, I1);
            D2 = min_i32(D1 + GEn, y1 + GOn);                                                 // :317
            I2 = hw::dpp_row_shl1(INF32, ish);                                                // :318-319 one diagonal down, infinity_ into the row's last diagonal
            if constexpr (TRACE) {
                const uint32_t tm = M2 & 3u, ti = I2 & 3u, td = D2 & 3u;
                M2 ^= tm; I2 = (I2 & ~3u) | 1u; D2 |= 3u;
                bw[u] = bpe | (tm | ti << 2 | td << 4) << 6 | fO >> 1;
            }
            // the operands move on: x + 1 becomes x; the row's last lane takes the tile's next new entry (at the row's first lane now: one rotation brings it over and leaves the
            // one after it there), every other lane its upper neighbour's; the read record comes from the lower neighbour, the row's first lane takes the next new one
            cE = nE; GO = GOn; GE = GEn;
            Fe.x = hw::dpp_row_rol1(Fe.x); Fe.y = hw::dpp_row_rol1(Fe.y);
            nE.x = hw::dpp_row_shl1(Fe.x, nE.x); nE.y = hw::dpp_row_shl1(Fe.y, nE.y);
            rw = hw::dpp_row_shr1(Fr, rw);
            Fr = hw::dpp_row_rol1(Fr);
        }
        if constexpr (TRACE) {                                                                // this lane's CH words of the tile: half of its 64-byte line
            uint4* dst = (uint4*)(bpt + (size_t)(k0 >> 4) * 1024 + (k0 & 15u));
            dst[0] = make_uint4(bw[0], bw[1], bw[2], bw[3]); dst[1] = make_uint4(bw[4], bw[5], bw[6], bw[7]);
        }
    };
    auto run_tile = [&](uint32_t k0, auto init_c, auto cap_c) __attribute__((always_inline)) {   // sixteen iterations; the next tile's new values in flight meanwhile
        const uint2 Pe = tab_at(k0 + 33u + li); const uint32_t Pr = rec_at((uint32_t)B + k0 + 17u + li);
        run_chunk(k0, init_c, cap_c); run_chunk(k0 + CH, init_c, cap_c);
        Fe = Pe; Fr = Pr;                                                                     // (sixteen rotations have brought Fe / Fr back to where they started: replaced whole)
    };
    const uint32_t k_cap0 = (t_lo & ~15u) > (uint32_t)B ? (t_lo & ~15u) : (uint32_t)B;       // no row has an end cell before iteration t_lo
    uint32_t k = 0;
    if (t_lo < (uint32_t)B) run_tile(0, BoolC<true>{}, BoolC<true>{}); else run_tile(0, BoolC<true>{}, BoolC<false>{});   // B = 16: the initialiser's iterations are the first tile
    for (k = 16; k < k_cap0 && k < KC; k += 16) run_tile(k, BoolC<false>{}, BoolC<false>{});
    for (; k < KC; k += 16) run_tile(k, BoolC<false>{}, BoolC<true>{});

    // first minimum over the row's end cells: per lane the candidates were visited in increasing diagonal order, so strict < kept the first
    uint32_t kv = have ? (best ^ 0x80000000u) : 0xffffffffu, ks = have ? best_s : 0xffffffffu;
    for (int m = 1; m < B; m <<= 1) {
        const uint32_t ov = hw::shfl_xor(kv, m), os = hw::shfl_xor(ks, m);
        if (ov < kv || (ov == kv && os < ks)) { kv = ov; ks = os; }
    }
    if (li == 0) {
        const bool none = kv == 0xffffffffu;                                                  // no end cell below infinity_: minscore stays infinity_, minscoreidx -1 (:269-270)
        const uint32_t biased = none ? (INF32 ^ 0x80000000u) : kv;
        const int32_t score = (int32_t)biased >> 2;
        if constexpr (TRACE) { TraceEnd e; e.score = score; e.sidx = none ? -1 : (int32_t)ks; p.ends[task] = e; }
        else if (t.pair != kPadTask) hw::atomic_min_i32(p.pair_best + t.pair, score);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Long reads at bands 128 / 256, int32 lanes (BASELINE.json configs[4]): ONE TASK PER WORKGROUP, its B band diagonals over B / 64 waves, one diagonal
// per lane. A long-read batch holds ~10^3 tasks of ~10^4 dependent iterations: with one wave per task (k_dp_wide) a SIMD holds one wave and the launch
// runs at the latency of a 4-diagonal dependency chain per lane; here the same batch puts four waves on every SIMD and a wave's chain is a quarter
// as long. Inside a wave D and I move between neighbouring lanes with a DPP shift as everywhere else; across the wave border they travel through
// 64-bit mailboxes in LDS, {value, iteration}, written by the border lane as soon as the value exists and read (re-read until the iteration tag
// matches) by the neighbour wave's border lane where it is needed:
//   dsh(k), the D hand-down, depends only on iteration k - 1: posted FIRST in iteration k, consumed after the M / I updates of the same iteration;
//   ish(k), the I hand-up, exists after the even diagonal: posted there, consumed at the end of the iteration.
// So neither wave waits in the steady state, no workgroup barrier is involved, and a wave can run at most one iteration ahead of its neighbours (ring of
// four slots per mailbox). Operands come from per-read record rows (DevBatch::rrecW) and the per-base tables in chunks of eight iterations, the next chunk in
// flight while this one computes; the fast-cost form (pure ACGT read on a clean haplotype) is perm + min + shift-add per cell like k_dp's.
// Same recurrence, same traceback words and layout as k_dp_wide<B, TRACE, false>: k_walk<B, 1, B / 64> serves both.
// ------------------------------------------------------------------------------------------------------------------
// PL = planes per wave: a wave owns PL runs of 64 adjacent diagonals (diagonal (wave * PL + c) * 64 + lane in plane c), a task B / (64 PL) waves.
//   PL = 1: four waves per task at band 256, borders through the mailboxes - the form for FEW tasks (a traceback launch of ~10^2 tasks, any batch below
//           ~10^3 tasks): it spreads a task over four SIMDs and quarters the dependency chain.
//   PL = B / 64: one wave per task, the planes' borders cross with a DPP rotate (lane 63 of plane c meets lane 0 of plane c + 1 inside the wave), no mailbox,
//           no scalar-unit work per iteration - the form for a launch that has a task for every SIMD: the planes give the lone wave four independent chains.
template <int B, int PL, bool TRACE, bool GENERIC>
OCT_MAX_THREADS(B / PL) OCT_KERNEL(k_dp_mw)(DpParams p)
{
    constexpr int C = B / 64, NW = C / PL;
    static_assert((B == 128 || B == 256) && (PL == 1 || PL == C), "multi-wave streaming kernel: bands 128 and 256, one plane or all planes per wave");
    __shared__ unsigned long long mbox_up[NW][4], mbox_dn[NW][4];        // [producer wave][iteration & 3]
    __shared__ uint32_t red_v[NW], red_s[NW];
    const uint32_t tid = hw::thread_idx(), lane = tid & 63, wave = hw::readfirstlane(tid >> 6);
    const DevTask* tasks = p.tasks; uint32_t n_tasks = p.n_tasks;
    if (p.ref.totals) { uint32_t first; task_list_range(p.ref, first, n_tasks); tasks += first; }   // device-sized launch: the grid is the host's bound
    const uint32_t task = hw::block_idx();
    if (task >= n_tasks) return;                                           // (whole workgroups)
    if constexpr (NW > 1) {
        if (lane < 4) { mbox_up[wave][lane] = ~0ull; mbox_dn[wave][lane] = ~0ull; }   // no iteration carries this tag
        hw::block_sync();
    }
    const DevTask t = tasks[task];
    const uint32_t ro = p.roff[t.read], T = p.roff[t.read + 1] - ro;
    const uint32_t K = T + B;
    constexpr int CH = PL == 1 ? 8 : 4;                                    // iterations per operand chunk
    const uint32_t KC = (K + CH - 1) / CH * CH;
    const uint32_t i0 = wave * PL * 64 + lane;                             // this lane's band diagonal in plane 0 (plane c: + 64 c)
    const uint2* tab = (p.rrev[t.read] ? p.tabR : p.tabF) + p.hoff[t.hap] + t.off;
    const uint32_t* rrow = p.rrecW + (size_t)t.read * p.rrec_stride;          // entry j = read position j - B: selector | base << 8 | quality << 24
    const uint32_t NUCW = (uint32_t)(int32_t)(int16_t)(p.nuc4 & 0xffffu);
    uint32_t M1[PL], I1[PL], D1[PL], M2[PL], I2[PL], D2[PL];
#pragma unroll
    for (int c = 0; c < PL; ++c) M1[c] = I1[c] = D1[c] = M2[c] = I2[c] = D2[c] = INF32;
    uint32_t best = INF32, best_s = 0; bool have = false;

    struct Chunk { uint2 e[PL][CH]; uint32_t r[PL][CH]; };
    // what iterations k0 .. k0 + CH - 1 newly need: table entry x + 1 = k + i + 1 and read record t = k - i. Two running pointers per lane, the loads at
    // constant offsets (no address arithmetic per load). The last chunks run a few entries past the task's window: the tables carry slack for that
    // (upload) and those iterations feed no end cell.
    const uint2* tab_next = tab + i0 + 1;
    const uint32_t* rec_next = rrow + ((uint32_t)B - i0);
    auto fetch = [&](Chunk& ch) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < PL; ++c) {
#pragma unroll
            for (int u = 0; u < CH; ++u) { ch.e[c][u] = tab_next[64 * c + u]; ch.r[c][u] = (rec_next - 64 * c)[u]; }
        }
        tab_next += CH; rec_next += CH;
    };
    auto cost = [&](const uint32_t rec, const uint2 a, uint32_t& flag) -> uint32_t {   // update_match_state (:121-132), already shifted by the trace bits
        if constexpr (GENERIC) {
            const uint32_t rc = (rec >> 8) & 0x1ffu, q4 = (rec >> 24) << 2;
            const uint32_t h = a.x & 0xffu, m = (a.x >> 8) & 0xffu, p4 = ((a.x >> 16) & 0xffu) << 2, isn = a.x >> 24;
            const uint32_t inner = rc == m ? p4 : q4;
            uint32_t c = min_i32(q4, inner);
            if (rc == h) c = 0;
            flag = rc != h ? 0x8000u : 0u;
            return min_i32(c, isn ? 8u : INF32);
        } else {
            const uint32_t cp = hw::perm(a.x, a.x, rec) & 0xffu;                   // this read base's cap at this haplotype position (0xff: none); only selector byte 0 counts
            const uint32_t q = rec >> 24;
            const uint32_t c = cp < q ? cp : q;                                    // min(quality, cap): one sub-dword-select min
            flag = c ? 0x8000u : 0u;                                               // the walk charges exactly c inside a flank
            return c << 2;
        }
    };
    // Mailboxes (NW > 1). A post is one 64-bit LDS store by the border lane (a store by every lane into private sink words, to spare the exec-mask round trip,
    // made the value visible later and cost the latency-bound traceback launch 5 %). A take is a broadcast read of the neighbour's mailbox by the whole wave,
    // repeated until it carries iteration k - a wave-uniform test (waves without that neighbour compare under a zero mask and never wait) - after which the
    // border lane selects the value.
    const bool first_lane = wave > 0 && lane == 0, last_lane = wave + 1 < (uint32_t)NW && lane == 63;   // this lane's D / I neighbour lives in another wave
    const unsigned long long* const take_up = &mbox_up[wave > 0 ? wave - 1 : 0][0];
    const unsigned long long* const take_dn = &mbox_dn[wave + 1 < (uint32_t)NW ? wave + 1 : wave][0];
    const uint32_t need_up = wave > 0 ? ~0u : 0u, need_dn = wave + 1 < (uint32_t)NW ? ~0u : 0u;
    auto take = [&](const unsigned long long* box, uint32_t need, uint32_t k) -> uint32_t {
        unsigned long long v = hw::lds_load_u64(box);
        while (((hw::readfirstlane((uint32_t)(v >> 32)) ^ k) & need) != 0) { hw::spin_pause(); v = hw::lds_load_u64(box); }
        return (uint32_t)v;
    };
    const uint32_t inf_lo = lane == 0 ? INF32 : 0u, inf_hi = lane == 63 ? INF32 : 0u;                   // infinity_ into the band's first / last diagonal
    uint32_t* bpt[PL];                                                     // this lane's 16-word line of plane c in tile 0 (+ tile * C * 1024 + (k & 15))
#pragma unroll
    for (int c = 0; c < PL; ++c) { const uint32_t i = i0 + 64 * c; bpt[c] = TRACE ? p.bp + (size_t)task * p.k_cap * C * 1024 + ((size_t)(i % C) * 64 + i / C) * 16 : nullptr; }

    uint2 cE[PL]; uint32_t GO[PL], GE[PL];                                 // table entry of x = k + i per plane, its gap penalties unpacked
#pragma unroll
    for (int c = 0; c < PL; ++c) { cE[c] = tab[i0 + 64 * c]; GO[c] = cE[c].y & 0xffffu; GE[c] = cE[c].y >> 16; }
    auto run_chunk = [&](uint32_t k0, const Chunk& cur, auto init_c, auto cap_c) __attribute__((always_inline)) {   // iterations k0 .. k0 + CH - 1 (k0 a multiple of CH); inlined at every call site: the state stays in registers
        constexpr bool INIT = decltype(init_c)::value, CAP = decltype(cap_c)::value;
        uint32_t bw[PL][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const uint32_t k = k0 + (uint32_t)u;
            uint32_t x2[PL], dsh[PL], ish[PL], bpe[PL], GOn[PL], GEn[PL], fO[PL];
            // the D hand-down of this iteration needs nothing of it: first, so that the next wave finds it early
#pragma unroll
            for (int c = 0; c < PL; ++c) {
                GOn[c] = cur.e[c][u].y & 0xffffu; GEn[c] = cur.e[c][u].y >> 16;
                if constexpr (INIT) { if (k == i0 + 64 * c) { M1[c] = NUL32; M2[c] = NUL32; } }   // rolling initialiser
                x2[c] = min_i32(M2[c], I2[c]);
                dsh[c] = min_i32(D2[c] + GEn[c], x2[c] + GOn[c]);                             // :293
            }
            if constexpr (NW > 1) { if (last_lane) hw::lds_store_u64(&mbox_up[wave][u & 3], (unsigned long long)k << 32 | dsh[PL - 1]); }
#pragma unroll
            for (int c = 0; c < PL; ++c) {
                uint32_t fE;
                const uint32_t m1 = min_i32(M1[c], min_i32(I1[c], D1[c]));                    // :284
                if constexpr (CAP) { if (k == T + i0 + 64 * c && (int32_t)m1 < (int32_t)best) { best = m1; best_s = 2 * k; have = true; } }   // :285-291
                M1[c] = m1 + cost(cur.r[c][u], cE[c], fE);                                    // :292
                I1[c] = min_i32(I2[c] + GE[c], M2[c] + GO[c]) + NUCW;                         // :295
                bpe[c] = 0;
                if constexpr (TRACE) { const uint32_t tm = M1[c] & 3u, ti = I1[c] & 3u; M1[c] ^= tm; I1[c] = (I1[c] & ~3u) | 1u; bpe[c] = tm | ti << 2 | fE; }
                ish[c] = min_i32(I1[c] + GE[c], M1[c] + GO[c]) + NUCW;                        // :318
            }
            if constexpr (NW > 1) { if (first_lane) hw::lds_store_u64(&mbox_dn[wave][u & 3], (unsigned long long)k << 32 | ish[0]); }
            // :294 one diagonal up: inside a plane a DPP shift; lane 0 takes lane 63 of the plane below (a DPP rotate of it is the fill of the shift), the
            // wave's first plane the neighbour wave's mailbox, the band's first diagonal infinity_
            D1[0] = hw::dpp_wave_shr1_z(dsh[0]) | inf_lo;
            if constexpr (NW > 1) { const uint32_t v = take(take_up + (u & 3), need_up, k); D1[0] = first_lane ? v : D1[0]; }
#pragma unroll
            for (int c = 1; c < PL; ++c) D1[c] = hw::dpp_wave_shr1(hw::dpp_wave_ror1(dsh[c - 1]), dsh[c]);
#pragma unroll
            for (int c = 0; c < PL; ++c) {
                if constexpr (TRACE) { const uint32_t td = D1[c] & 3u; D1[c] |= 3u; bpe[c] |= td << 4; }
                const uint32_t m2 = min_i32(x2[c], D2[c]);                                    // :308
                if constexpr (CAP) { if (k == T + i0 + 64 * c && (int32_t)m2 < (int32_t)best) { best = m2; best_s = 2 * k + 1; have = true; } }
                M2[c] = m2 + cost(cur.r[c][u], cur.e[c][u], fO[c]);                           // :316
                const uint32_t y1 = min_i32(M1[c], I1[c]);
                D2[c] = min_i32(D1[c] + GEn[c], y1 + GOn[c]);                                 // :317
            }
            // :318-319 one diagonal down, the mirror image
            I2[PL - 1] = hw::dpp_wave_shl1_z(ish[PL - 1]) | inf_hi;
            if constexpr (NW > 1) { const uint32_t v = take(take_dn + (u & 3), need_dn, k); I2[PL - 1] = last_lane ? v : I2[PL - 1]; }
#pragma unroll
            for (int c = 0; c + 1 < PL; ++c) I2[c] = hw::dpp_wave_shl1(hw::dpp_wave_rol1(ish[c + 1]), ish[c]);
#pragma unroll
            for (int c = 0; c < PL; ++c) {
                if constexpr (TRACE) {
                    const uint32_t tm = M2[c] & 3u, ti = I2[c] & 3u, td = D2[c] & 3u;
                    M2[c] ^= tm; I2[c] = (I2[c] & ~3u) | 1u; D2[c] |= 3u;
                    bw[c][u] = bpe[c] | (tm | ti << 2 | td << 4) << 6 | fO[c] >> 1;
                }
                cE[c] = cur.e[c][u]; GO[c] = GOn[c]; GE[c] = GEn[c];
            }
        }
        if constexpr (TRACE) {                                                               // this lane's CH words of the tile per plane: a piece of its 64-byte line
#pragma unroll
            for (int c = 0; c < PL; ++c) {
                uint4* dst = (uint4*)(bpt[c] + (size_t)(k0 >> 4) * C * 1024 + (k0 & 15u));
                dst[0] = make_uint4(bw[c][0], bw[c][1], bw[c][2], bw[c][3]);
                if constexpr (CH == 8) dst[1] = make_uint4(bw[c][4], bw[c][5], bw[c][6], bw[c][7]);
            }
        }
    };
    // two operand chunks, used in turn: the one not being consumed is in flight (no register copies between them)
    Chunk ca, cb;
    fetch(ca);
    bool flip = false;
    auto advance = [&](uint32_t k0, auto init_c, auto cap_c) __attribute__((always_inline)) {
        const bool more = k0 + CH < KC;
        if (!flip) { if (more) fetch(cb); run_chunk(k0, ca, init_c, cap_c); }
        else       { if (more) fetch(ca); run_chunk(k0, cb, init_c, cap_c); }
        flip = !flip;
    };
    const uint32_t t_lo = (T / CH * CH) > (uint32_t)B ? (T / CH * CH) : (uint32_t)B;         // no end cell before iteration T
    uint32_t k = 0;
    if (T < (uint32_t)B) { for (; k < (uint32_t)B; k += CH) advance(k, BoolC<true>{}, BoolC<true>{}); }
    for (; k < (uint32_t)B; k += CH) advance(k, BoolC<true>{}, BoolC<false>{});
    for (; k < t_lo; k += CH) advance(k, BoolC<false>{}, BoolC<false>{});
    for (; k < KC; k += CH) advance(k, BoolC<false>{}, BoolC<true>{});

    // first minimum over the end cells: per lane the candidates came in increasing diagonal order WITHIN a plane; across planes, lanes and waves by (value, diagonal index)
    uint32_t kv = have ? (best ^ 0x80000000u) : 0xffffffffu, ks = have ? best_s : 0xffffffffu;
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t ov = hw::shfl_xor(kv, m), os = hw::shfl_xor(ks, m);
        if (ov < kv || (ov == kv && os < ks)) { kv = ov; ks = os; }
    }
    if constexpr (NW > 1) {
        if (lane == 0) { red_v[wave] = kv; red_s[wave] = ks; }
        hw::block_sync();
    }
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) { const uint32_t ov = red_v[w], os = red_s[w]; if (ov < kv || (ov == kv && os < ks)) { kv = ov; ks = os; } }
        const bool none = kv == 0xffffffffu;                                                  // no end cell below infinity_: minscore stays infinity_, minscoreidx -1 (:269-270)
        const uint32_t biased = none ? (INF32 ^ 0x80000000u) : kv;
        const int32_t score = (int32_t)biased >> 2;
        if constexpr (TRACE) { TraceEnd e; e.score = score; e.sidx = none ? -1 : (int32_t)ks; p.ends[task] = e; }
        else if (t.pair != kPadTask) hw::atomic_min_i32(p.pair_best + t.pair, score);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// traceback walk + flank score
// ------------------------------------------------------------------------------------------------------------------
template <int B, int TPR, int C>
OCT_KERNEL(k_walk_strings)(WalkParams w)   // test seam only: emits the gapped strings (set_alignments :165-231), one step per iteration
{
    constexpr uint32_t ROWS = 64 * C / B, G = TPR * ROWS;   // C > 1: band 64 x C on one wave, diagonal i lives in lane i / C, plane i % C
    const uint32_t ti = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (ti >= w.n_tasks) return;
    const DevTask t = w.tasks[ti];
    if (t.pair == kPadTask) return;
    const TraceEnd end = w.ends[ti];
    const uint32_t group = ti / G, slot = ti % G, row = slot / TPR, half = slot % TPR;
    const uint32_t ro = w.roff[t.read]; const int32_t T = (int32_t)(w.roff[t.read + 1] - ro);
    const uint32_t ho = w.hoff[t.hap]; const int32_t Lh = (int32_t)(w.hoff[t.hap + 1] - ho);
    const int32_t L = T + 2 * B - 1, off = (int32_t)t.off;
    const uint8_t* target = w.rbases + ro; const int8_t* quals = (const int8_t*)w.rquals + ro;
    const bool fwd = !w.rrev[t.read];
    const uint8_t* truth = w.hbases + ho + off;
    const uint8_t* mask = (fwd ? w.maskF : w.maskR) + ho + off; const int8_t* prior = (fwd ? w.priorF : w.priorR) + ho + off;
    const int8_t* go = w.go + ho + off; const int8_t* ge = w.ge + ho + off;
    const bool seam = w.out_first_pos != nullptr;
    // flank sizes in window coordinates (pair_hmm.hpp:572-587)
    int32_t lhs = 0, rhs = 0; bool want_flank = true;
    if (seam) {
        want_flank = w.seam_lhs != nullptr;
        if (want_flank) { lhs = w.seam_lhs[ti]; rhs = w.seam_rhs[ti]; }
    } else {
        const uint32_t g = w.hap_region[t.hap];
        lhs = (int32_t)w.reg_lhs[g];
        if (lhs < off) lhs = 0; else { lhs -= off; if (lhs < 0) lhs = 0; }
        rhs = (int32_t)w.reg_rhs[g];
        if (off + L < Lh - rhs) rhs = 0; else { rhs += off + L; rhs -= Lh; if (rhs < 0) rhs = 0; }
    }
    const int32_t rhs_begin = L - rhs;
    const int32_t n_diag = 2 * (T + B) + 1; const int64_t n_flat = (int64_t)n_diag * B;
    // backpointers of this group: k_cap tiles of [64 lanes][16 iterations] dwords; one 64-byte line = 16 consecutive iterations of one lane
    const uint4* bpg = (const uint4*)(w.bp + (size_t)group * w.k_cap * C * 1024);
    uint32_t line_id = 0xffffffffu; uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
    auto bits_at = [&](int64_t flat) -> uint32_t {          // 6 backpointer bits of band cell `flat` = diagonal * B + lane
        const int32_t s = (int32_t)(flat / B), i = (int32_t)(flat % B);
        if (s >= 2 * (T + B)) return 0;                     // last row of the reference's array is never written (zeros)
        const uint32_t k = (uint32_t)s >> 1;
        const uint32_t id = C == 1 ? (k >> 4) * 64 + row * B + (uint32_t)i : ((k >> 4) * C + (uint32_t)i % C) * 64 + (uint32_t)i / C;
        if (id != line_id) { const uint4* l = bpg + (size_t)id * 4; c0 = l[0]; c1 = l[1]; c2 = l[2]; c3 = l[3]; line_id = id; }
        const uint4 q = (k & 8) ? ((k & 4) ? c3 : c2) : ((k & 4) ? c1 : c0);
        const uint32_t wv = (k & 2) ? ((k & 1) ? q.w : q.z) : ((k & 1) ? q.y : q.x);
        return (wv >> (16 * half + 6 * (s & 1))) & 63u;
    };
    char* a1 = seam ? w.out_align1 + w.out_align_off[ti] : nullptr;
    char* a2 = seam ? w.out_align2 + w.out_align_off[ti] : nullptr;

    int32_t first_pos = 0, flank = 0, msz = 0, alnidx = 0;
    bool ok = true;
    int32_t sidx = end.sidx;
    if (sidx < 0) ok = false;                               // minscore never updated (:176-179)
    int32_t i = sidx / 2 - T, y = T, x = sidx - y;
    if (ok) { const int64_t f0 = (int64_t)sidx * B + i; if (f0 < 0 || f0 >= n_flat) ok = false; }   // :186-190
    if (ok) {
        uint32_t state = bits_at((int64_t)sidx * B + i) & 3u;   // :191
        sidx -= 2;
        while (y > 0) {                                     // :194
            if (sidx < 0 || i < 0) { ok = false; break; }   // :195-199
            const int64_t f = (int64_t)sidx * B + i;
            if (f >= n_flat) { ok = false; break; }         // the reference would read past its array here (UB)
            const uint32_t bits = bits_at(f);
            const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;   // :200 (our D bits sit at 4-5)
            if (state == 0) {                               // match :201-204
                sidx -= 2; --x; --y;
                if (seam) { a1[alnidx] = (char)truth[x]; a2[alnidx] = (char)target[y]; }
                if (want_flank && (x < lhs || x >= rhs_begin)) {                      // calculate_flank_score_helper :383-397
                    const uint32_t hc = truth[x], rc = target[y];
                    if (hc != rc) {
                        if (hc != 'N') { int32_t q = quals[y]; if (mask[x] == rc && prior[x] < q) q = prior[x]; flank += q; }
                        else flank += 2;
                    }
                    ++msz;
                }
            } else if (state == 1) {                        // insert :205-209
                i += sidx & 1; sidx -= 1; --y;
                if (seam) { a1[alnidx] = '-'; a2[alnidx] = (char)target[y]; }
                if (want_flank && (x < lhs || x >= rhs_begin)) {                      // :399-411
                    const int32_t gi = x - 1 < 0 ? 0 : x - 1;                         // x-1 == -1 is out of bounds in the reference (UB): clamp
                    const bool prev_ins = y != 0 && new_state == 1;                   // first alignment column has prev_state = match (:369)
                    flank += (prev_ins ? ge[gi] : go[gi]) + w.nuc_prior;
                    ++msz;
                }
            } else {                                        // delete :210-215
                sidx -= 1; i -= sidx & 1; --x;
                if (seam) { a1[alnidx] = (char)truth[x]; a2[alnidx] = '-'; }
                if (want_flank && (x < lhs || x >= rhs_begin)) flank += (new_state == 3 ? ge[x] : go[x]);   // :413-424
            }
            state = new_state;
            ++alnidx;
        }
        first_pos = x;
    }
    if (!ok) first_pos = -1;
    if (seam) {                                             // first_pos / flank score / mask size are reported by k_walk
        if (ok) {
            a1[alnidx] = 0; a2[alnidx] = 0;
            for (int32_t a = 0, b = alnidx - 1; a < b; ++a, --b) {   // :223-230
                char c = a1[a]; a1[a] = a1[b]; a1[b] = c; c = a2[a]; a2[a] = a2[b]; a2[b] = c;
            }
        }
        (void)first_pos;
        return;
    }
    if (!ok) return;                                        // lowest(): contributes nothing to the max (pair_hmm.hpp:750-752)
    if (T - msz < 2) flank = 0;                             // :757-759
    const int32_t score = end.score;
    const int32_t pen = flank <= score ? score - flank : flank + score;   // :760-764
    hw::atomic_min_i32(w.pair_best + t.pair, pen);
}

// Penalty of one in-flank alignment column (calculate_flank_score_helper, simd_pair_hmm.hpp:383-424). Deliberately NOT inlined: the
// unrolled sweep of k_walk has 32 call sites on its rare event-buffer-overflow path and must stay inside the instruction cache.
struct WalkPricing {
    const uint8_t* rbases; const int8_t* rquals; const uint8_t* hbases; const uint8_t* mask; const int8_t* prior;
    const int8_t* go; const int8_t* ge;       // rbases/rquals at the read's first base, the haplotype arrays at the window's first base
};
// A queued event is ONE word: kind << 30 | (x - y) << 20 | x. A match column (kind 0) needs both coordinates, and inside the band x - y is the band diagonal,
// 0 ... 2B - 1 (x + y = sidx + 2 and band lane i = (x - y) / 2 all along the walk): ten bits for it leave twenty for x, i.e. reads of up to a million bases (rounds 1-5 held
// x and y in 15 bits each and refused reads from 32 k bases on). Gap columns (kinds 1, 2) are priced at x alone: their callers pass y = x. A walk that has left the band through
// the reference's flat-index rule can have any x - y: its column does not fit and is priced at once instead of queued (walk_price_lo, the one non-inlined pricing function;
// two arguments and one call site per step - a third argument cost k_walk four registers and a wave per SIMD).
constexpr uint32_t kWalkEventXBits = 20, kWalkEventDBits = 10;
constexpr uint32_t kMaxWindowBases = 1u << kWalkEventXBits;              // T + 2B must stay below this (host_upload.hh)
OCT_DEVICE_NOINLINE int32_t walk_price_lo(WalkPricing p, uint32_t lo, int32_t ey)      // lo = kind << 30 | x
{
    const uint32_t kind = lo >> 30, ex = lo & 0x3fffffffu;
    if (kind == 0) {
        const uint32_t hc = p.hbases[ex], rc = p.rbases[ey];
        if (hc == rc) return 0;
        if (hc == 'N') return 2;
        int32_t q = p.rquals[ey];
        const int32_t pr = p.prior[ex];
        if (p.mask[ex] == rc && pr < q) q = pr;
        return q;
    }
    return (kind == 1 ? p.go : p.ge)[ex];
}
OCT_DEVICE int32_t walk_price_event(WalkPricing p, uint32_t e)
{
    const uint32_t ex = e & ((1u << kWalkEventXBits) - 1u);
    return walk_price_lo(p, (e & 0xc0000000u) | ex, (int32_t)(ex - ((e >> kWalkEventXBits) & ((1u << kWalkEventDBits) - 1u))));
}
// queue the column (kind, x, y) - gap columns: y = x - or price it now: `room` = the queue has a free slot
#define OCT_WALK_EVENT(kind, ex, ey, room, STORE)                                                                     \
    do {                                                                                                            \
        const uint32_t lo_ = (uint32_t)(kind) << 30 | (uint32_t)(ex), d_ = (uint32_t)((ex) - (ey));                 \
        if ((room) && d_ < (1u << kWalkEventDBits)) { const uint32_t e_ = lo_ | d_ << kWalkEventXBits; STORE; }     \
        else flank += walk_price_lo(pricing, lo_, (ey));                                                            \
    } while (0)

// Production walk: one thread per traceback task, all 64 tasks of a wave sweep the band iterations k from the top tile down IN
// LOCKSTEP. Per 16-iteration tile every lane holds its own 64-byte backpointer line in registers (statically indexed in the
// unrolled sweep), so a tile costs one batch of line loads for the whole wave instead of a memory round trip per step. In-flank
// penalties are not loaded while walking: the DP left a "this match cell costs something" flag next to the labels, gap columns are
// rare, and both are queued as events in LDS and priced in a second uniform loop.
constexpr uint32_t kWalkEvents = 12;

// STAGE (region-sized launches, one wave per workgroup): a launch of ~100 waves has nothing to hide a memory round trip behind, and every change of
// band lane (an indel column) in any of a wave's 64 walks used to stall the whole lockstep sweep for a line fetch (half of the old kernel's time on a
// 300 x 24 region). Here the wave copies the current tile of all its task groups into LDS first (64 / G blocks of 4 KB, coalesced) and every step
// reads its word from there: a lane change is just another LDS address. Task row r keeps its B lines at stride 17 words, rows B * 17 + 1 words apart:
// walkers on the same band lane and iteration (the usual case) hit 32 different banks.
constexpr uint32_t walk_stage_row_words(uint32_t B) { return B * 17 + 1; }
inline size_t walk_stage_lds_bytes(uint32_t B, uint32_t tpr) { return (64 * kWalkEvents + (64 / tpr) * walk_stage_row_words(B)) * sizeof(uint32_t); }

template <int B, int TPR, int C, bool STAGE = false>
OCT_MAX_THREADS(STAGE ? 64 : 256) OCT_KERNEL(k_walk)(WalkParams w)
{
    chain_head_priority<true>();
    constexpr uint32_t ROWS = 64 * C / B, G = TPR * ROWS;   // C > 1: band 64 x C on one wave, diagonal i lives in lane i / C, plane i % C
    static_assert(!STAGE || C == 1, "staged walk: bands up to 64");
    OCT_DYN_SMEM(smem);
    uint32_t* evbuf = (uint32_t*)smem + hw::thread_idx() * kWalkEvents;
    uint32_t* tbuf = (uint32_t*)smem + 64 * kWalkEvents;                  // STAGE: [64 / TPR task rows][B lines x 17 words + 1]
    constexpr uint32_t RS = walk_stage_row_words(B);
    const uint32_t ti = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    const DevTask* tasks = w.tasks; uint32_t n_tasks = w.n_tasks;
    if (w.ref.totals) {                                                   // device-sized launch: the grid is the host's bound, surplus waves leave here
        uint32_t first; task_list_range(w.ref, first, n_tasks); tasks += first;
        if ((ti & ~63u) >= n_tasks) return;
    }
    DevTask t; t.pair = kPadTask; t.read = 0; t.hap = 0; t.off = 0;
    if (ti < n_tasks) t = tasks[ti];
    const bool active = t.pair != kPadTask;
    TraceEnd end; end.score = 0; end.sidx = -1;
    if (active) end = w.ends[ti];
    const uint32_t group = ti / G, slot = ti % G, row = slot / TPR, half = slot % TPR;
    const uint32_t ro = w.roff[t.read]; const int32_t T = active ? (int32_t)(w.roff[t.read + 1] - ro) : 1;
    const uint32_t ho = w.hoff[t.hap]; const int32_t Lh = (int32_t)(w.hoff[t.hap + 1] - ho);
    const int32_t L = T + 2 * B - 1, off = (int32_t)t.off;
    const bool seam = w.out_first_pos != nullptr;
    int32_t lhs = 0, rhs = 0; bool want_flank = true;                     // flank sizes in window coordinates (pair_hmm.hpp:572-587)
    if (seam) {
        want_flank = w.seam_lhs != nullptr;
        if (want_flank && active) { lhs = w.seam_lhs[ti]; rhs = w.seam_rhs[ti]; }
    } else if (active) {
        const uint32_t g = w.hap_region[t.hap];
        lhs = (int32_t)w.reg_lhs[g];
        if (lhs < off) lhs = 0; else { lhs -= off; if (lhs < 0) lhs = 0; }
        rhs = (int32_t)w.reg_rhs[g];
        if (off + L < Lh - rhs) rhs = 0; else { rhs += off + L; rhs -= Lh; if (rhs < 0) rhs = 0; }
    }
    const int32_t rhs_begin = L - rhs;
    const int32_t n_diag = 2 * (T + B) + 1; const int64_t n_flat = (int64_t)n_diag * B;
    const uint4* bpg = (const uint4*)(w.bp + (size_t)group * w.k_cap * C * 1024);
    const uint32_t hshift = 16 * half;

    // walker state (set_alignments :180-193)
    int32_t sidx = end.sidx, i = sidx / 2 - T, y = T, x = sidx - T;
    int32_t flank = 0, msz = 0; uint32_t nev = 0, state = 0;
    // walker flags in ONE register word (separate bools ended up in scratch memory: the compiler merged their stores through a pointer select)
    constexpr uint32_t kOk = 1u, kFin = 2u;
    uint32_t fl = (active && sidx >= 0) ? kOk : kFin;
    if (fl & kOk) { const int64_t f0 = (int64_t)sidx * B + i; if (f0 < 0 || f0 >= n_flat) fl = kFin; }   // :186-190

    WalkPricing pricing;
    {
        const bool fwd = !w.rrev[t.read];
        const size_t hb0 = (size_t)ho + (uint32_t)off;
        pricing.rbases = w.rbases + ro; pricing.rquals = (const int8_t*)w.rquals + ro; pricing.hbases = w.hbases + hb0;
        pricing.mask = (fwd ? w.maskF : w.maskR) + hb0; pricing.prior = (fwd ? w.priorF : w.priorR) + hb0;
        pricing.go = w.go + hb0; pricing.ge = w.ge + hb0;
    }
    auto price_event = [&](uint32_t e) { flank += walk_price_event(pricing, e); };
    auto push_event = [&](uint32_t kind, int32_t ex, int32_t ey) {
        OCT_WALK_EVENT(kind, ex, ey, nev < kWalkEvents, evbuf[nev++] = e_);
    };
    // one alignment column from backpointer word `wv` of cell (sidx, i). Written with selects instead of a three-way branch (the
    // unrolled sweep below instantiates it 32 times per tile; the branchy form overflowed the instruction cache); only the rare
    // in-flank event push is a branch.
    // nothing is left to charge once a walk without left flank has passed the right one (early_stop: the traceback cannot fail when no lane wraps)
    const int32_t stop_below_x = (w.early_stop && lhs == 0) ? rhs_begin : INT32_MIN;
    auto step = [&](uint32_t wv) {
        const uint32_t par = (uint32_t)sidx & 1u;
        const uint32_t bits = (wv >> (hshift + 6 * par)) & 63u, mism = (wv >> (hshift + 15 - par)) & 1u;
        const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;                 // :200
        const bool isM = state == 0, isI = state == 1, isD = !isM && !isI;
        i += isI ? (sidx & 1) : 0;                                                              // insert :205-209
        sidx -= isM ? 2 : 1;                                                                    // match :201-204
        i -= isD ? (sidx & 1) : 0;                                                              // delete :210-215
        x -= isI ? 0 : 1; y -= isD ? 0 : 1;
        const bool in_flank = want_flank && (x < lhs || x >= rhs_begin);                        // calculate_flank_score_helper :383-424
        msz += (in_flank && !isD) ? 1 : 0;
        flank += (in_flank && isI) ? w.nuc_prior : 0;
        if (in_flank && (!isM || mism)) {
            const bool ext = isI ? (y != 0 && new_state == 1) : new_state == 3;                 // first alignment column has prev_state = match (:369)
            const int32_t xi = x - 1 < 0 ? 0 : x - 1;                                           // x-1 == -1 is out of bounds in the reference (UB): clamp
            const int32_t ex = isI ? xi : x;
            push_event(isM ? 0u : (ext ? 2u : 1u), ex, isM ? y : ex);                              // (a gap column is priced at x alone: y = x)
        }
        state = new_state;
        fl |= (y <= 0 || x < stop_below_x) ? kFin : 0u;                                         // :194 / early stop
    };
    auto slow_word = [&](int64_t flat) -> uint32_t {                                            // any cell by flat index = diagonal * B + lane
        const int32_t s = (int32_t)(flat / B), li = (int32_t)(flat % B);
        if (s >= 2 * (T + B)) return 0;                                                         // last row of the reference's array is never written
        const uint32_t k = (uint32_t)s >> 1;
        const size_t line = C == 1 ? (size_t)(k >> 4) * 64 + row * B + (uint32_t)li : ((size_t)(k >> 4) * C + (uint32_t)li % C) * 64 + (uint32_t)li / C;
        return w.bp[(size_t)group * w.k_cap * C * 1024 + line * 16 + (k & 15)];
    };

    // The walk's first move only reads the end cell's own label (:191-192). It is taken here, with one word fetched straight from the tiles, so that the
    // sweep below has ONE kind of step and no "started yet?" branch in it.
    if (fl & kOk) {
        const int64_t f0 = (int64_t)sidx * B + i;
        const uint32_t wv = slow_word(f0);
        state = (wv >> (hshift + 6 * ((uint32_t)sidx & 1u))) & 3u;
        sidx -= 2;
    }
    uint32_t kmax = ((fl & kOk) && sidx >= 0) ? (uint32_t)(sidx >> 1) : 0;
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = hw::shfl_xor(kmax, m); kmax = o > kmax ? o : kmax; }
    kmax = hw::readfirstlane(kmax);
    for (int32_t kt = (int32_t)(kmax >> 4); kt >= 0; --kt) {
        if (hw::ballot(!(fl & kFin)) == 0) break;                                              // every walk of the wave is over (early stops)
        uint32_t c[16];
        int32_t line_i = INT32_MIN;                                                             // no band lane: a walker outside the band takes the rare path below
        if constexpr (STAGE) {
            constexpr uint32_t NG = 64 / G;                                                     // task groups of this wave
            const uint32_t lane = hw::thread_idx() & 63u, group0 = (ti & ~63u) / G, n_groups = n_tasks / G;
            hw::wave_lds_fence();                                                               // the previous tile's reads are done
            constexpr uint32_t GB = NG < 8 ? NG : 8;                                            // groups per batch: up to 32 loads per lane in flight (a one-wave workgroup has the registers)
            for (uint32_t gl0 = 0; gl0 < NG; gl0 += GB) {
                uint4 v[GB][4];
#pragma unroll
                for (uint32_t a = 0; a < GB; ++a) {
                    const uint32_t gl = gl0 + a;
                    const bool in = gl < NG && group0 + gl < n_groups;
                    const uint4* src = (const uint4*)(w.bp + ((size_t)(group0 + (in ? gl : 0)) * w.k_cap + (uint32_t)kt) * 1024);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) v[a][j] = in ? src[lane + 64 * j] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (uint32_t a = 0; a < GB; ++a) {
                    const uint32_t gl = gl0 + a;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t u = lane + 64 * j, line = u >> 2, q = u & 3u;            // line = row * B + band lane, q = which four iterations
                        uint32_t* dst = tbuf + (gl * ROWS + line / B) * RS + (line % B) * 17 + 4 * q;
                        dst[0] = v[a][j].x; dst[1] = v[a][j].y; dst[2] = v[a][j].z; dst[3] = v[a][j].w;
                    }
                }
            }
            hw::wave_lds_fence();
        }
        const uint32_t* my_row = tbuf + ((hw::thread_idx() & 63u) / TPR) * RS;                  // STAGE: this walk's task row
        auto load_line = [&]() {
            const size_t line = C == 1 ? (size_t)kt * 64 + row * B + (uint32_t)i : ((size_t)kt * C + (uint32_t)i % C) * 64 + (uint32_t)i / C;
            const uint4* l = bpg + line * 4;
            const uint4 q0 = l[0], q1 = l[1], q2 = l[2], q3 = l[3];      // (plain loads: the two tasks of a word pair share these lines - non-temporal loads cost the walk 1.00 -> 1.29 ms, profiles/r06_s12b)
            c[0] = q0.x; c[1] = q0.y; c[2] = q0.z; c[3] = q0.w; c[4] = q1.x; c[5] = q1.y; c[6] = q1.z; c[7] = q1.w;
            c[8] = q2.x; c[9] = q2.y; c[10] = q2.z; c[11] = q2.w; c[12] = q3.x; c[13] = q3.y; c[14] = q3.z; c[15] = q3.w;
            line_i = i;
        };
        if constexpr (!STAGE) { if (!(fl & kFin) && i >= 0 && i < B) load_line(); else { for (int q = 0; q < 16; ++q) c[q] = 0; } }
#pragma nounroll
        for (int kk = 15; kk >= 0; --kk) {                                                      // kk is wave-uniform: c[kk] is an indexed register read (M0), and
            const int32_t k = kt * 16 + kk;                                                     // the loop body stays small enough for the instruction cache
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {                                                 // an insertion/deletion can add a second step at the same k
                if (!(fl & kFin) && (sidx >> 1) == k) {                                         // (a negative sidx never equals k >= 0)
                    uint32_t wv = 0;
                    bool go = true;
                    if constexpr (STAGE) {
                        if (i >= 0 && i < B) wv = my_row[(uint32_t)i * 17 + (uint32_t)kk];
                        else if (i < 0) { fl = (fl & ~kOk) | kFin; go = false; }                // :195-199
                        else {                                                                  // the reference indexes its array flat: lane overflow reads the next diagonal
                            const int64_t f = (int64_t)sidx * B + i;
                            if (f >= n_flat) { fl = (fl & ~kOk) | kFin; go = false; } else wv = slow_word(f);
                        }
                    } else {
                    wv = c[kk];
                    if (i != line_i) {                                                          // rare: the walk changed its band lane (an indel), or left the band
                        if (i < 0) { fl = (fl & ~kOk) | kFin; go = false; }                     // :195-199
                        else if (i >= B) {                                                      // the reference indexes its array flat: lane overflow reads the next diagonal
                            const int64_t f = (int64_t)sidx * B + i;
                            if (f >= n_flat) { fl = (fl & ~kOk) | kFin; go = false; } else wv = slow_word(f);
                        } else { load_line(); wv = c[kk]; }
                    }
                    }
                    if (go) step(wv);
                }
            }
        }
    }
    if (!(fl & kFin)) fl &= ~kOk;                                                               // ran off the first diagonal with target bases left (:195-199)
    const bool ok = (fl & kOk) != 0;
    const int32_t first_pos = ok ? x : -1;
    if (ok) for (uint32_t e = 0; e < nev; ++e) price_event(evbuf[e]);
    if (seam) {
        if (active) {
            w.out_first_pos[ti] = first_pos;
            if (want_flank) { w.out_flank[ti] = ok ? flank : 0; w.out_mask_size[ti] = ok ? msz : 0; }
        }
        return;
    }
    if (!ok) {
        if (w.pair_key && active) { w.task_key[ti] = ~0ull; hw::atomic_or_u32(w.err_flags, 1u); }   // simd_align throws HMMOverflow (:811-813)
        return;                                             // populate: lowest(), contributes nothing to the max (pair_hmm.hpp:750-752)
    }
    if (T - msz < 2) flank = 0;                             // :757-759 / :664-665
    const int32_t score = end.score;
    const int32_t pen = flank <= score ? score - flank : flank + score;   // :760-764 / :666-670
    if (w.pair_key) {                                       // align mode: compete for the pair under the reference's tie rules
        const uint32_t p = t.off + (uint32_t)B;             // in-range positions are >= B, so alignment_offset = position - B
        const uint32_t* P = w.pos + (size_t)t.pair * (uint32_t)w.max_pos; const uint32_t np = w.npos[t.pair];
        unsigned long long order = 0;                       // not in the mapped list: the original (or shifted original) position
        for (uint32_t j = 0; j < np; ++j) if (P[j] == p) { order = j + 1; break; }
        const unsigned long long key = (unsigned long long)(uint32_t)pen << 32 | order << 8;
        w.task_key[ti] = key;
        hw::atomic_min_u64(w.pair_key + t.pair, key);
        return;
    }
    hw::atomic_min_i32(w.pair_best + t.pair, pen);
}

// Region-sized launches (a few thousand walks of ~150 columns): the lockstep walker above puts them into ~100 waves that each pay ~32 step slots per tile for
// 64 DIFFERENT walks. Here ONE WALK OWNS A 16-LANE ROW (four walks per wave, ~1,600 waves for a 300 x 24 region):
//  * the row stages the whole band of its task row for the next K tiles in LDS (lane l copies the 64-byte lines of band lane l: one memory round trip per
//    K x 16 iterations, and a change of band lane - an indel column - is just another LDS address);
//  * a run of match columns walks straight down one band lane at one diagonal parity, i.e. along ONE staged line: its sixteen words are looked at by the
//    row's sixteen lanes at once (a ballot over "this word's label is not match") and the run is taken in ONE move, flank bookkeeping included - the
//    columns inside a flank are counted, and their "costs something" marks queued, by the lanes that hold them;
//  * everything else - the columns around an indel, a junk alignment that zig-zags through the band - goes one column at a time in an inner loop without
//    any cross-lane operation, until the next run starts.
// Every cross-lane operation sits in wave-uniform control flow (rows that are done idle along). Same steps, same events, same result as k_walk.
constexpr uint32_t kWalkRowEvents = 64;
#ifndef OCT_WALK_PREFETCH
#define OCT_WALK_PREFETCH 1
#endif
constexpr bool kPrefetch = OCT_WALK_PREFETCH != 0;
constexpr uint32_t kWalkRowLine = 20;                                      // words per staged line: 16 + pad (16-byte aligned; the 16 lanes of a row write their lines with b128 stores, two-way conflicts at most)
constexpr uint32_t walk_rows_tiles(uint32_t B) { return B <= 16 ? 4 : (B == 32 ? 2 : 1); }
constexpr uint32_t walk_rows_row_words(uint32_t B) { return kWalkRowEvents + walk_rows_tiles(B) * B * kWalkRowLine + 8; }   // (+ 8: the four rows of a wave start in different banks)
inline size_t walk_rows_lds_bytes(uint32_t B, uint32_t threads) { return threads / 16 * walk_rows_row_words(B) * sizeof(uint32_t); }

template <int B, int TPR>
OCT_MAX_THREADS(256) OCT_KERNEL(k_walk_rows)(WalkParams w)
{
    chain_head_priority<true>();
    constexpr uint32_t ROWS = 64 / B, G = TPR * ROWS;
    constexpr uint32_t K = walk_rows_tiles(B), LPL = B <= 16 ? 1 : B / 16, LS = kWalkRowLine;   // tiles staged at a time, band lines per lane and tile
    static_assert(B <= 64, "one band row per task group row");
    OCT_DYN_SMEM(smem);
    const uint32_t lane = hw::thread_idx() & 63u, l16 = lane & 15u, rowbase = lane & 48u;
    uint32_t* evbuf = (uint32_t*)smem + (hw::thread_idx() >> 4) * walk_rows_row_words(B);      // [kWalkRowEvents] queued in-flank events of this row's walk
    uint32_t* rowt = evbuf + kWalkRowEvents;                                                    // [K tiles][B band lanes][LS] staged backpointer lines
    const uint32_t ti = (hw::block_idx() * hw::block_dim() + hw::thread_idx()) >> 4;
    const DevTask* tasks = w.tasks; uint32_t n_tasks = w.n_tasks;
    if (w.ref.totals) {                                                   // device-sized launch: the grid is the host's bound, surplus waves leave here
        uint32_t first; task_list_range(w.ref, first, n_tasks); tasks += first;
        if ((ti & ~3u) >= n_tasks) return;
    }
    DevTask t; t.pair = kPadTask; t.read = 0; t.hap = 0; t.off = 0;
    if (ti < n_tasks) t = tasks[ti];
    const bool active = t.pair != kPadTask;
    TraceEnd end; end.score = 0; end.sidx = -1;
    if (active) end = w.ends[ti];
    const uint32_t group = ti / G, slot = ti % G, row = slot / TPR, half = slot % TPR;
    const uint32_t ro = w.roff[t.read]; const int32_t T = active ? (int32_t)(w.roff[t.read + 1] - ro) : 1;
    const uint32_t ho = w.hoff[t.hap]; const int32_t Lh = (int32_t)(w.hoff[t.hap + 1] - ho);
    const int32_t L = T + 2 * B - 1, off = (int32_t)t.off;
    const bool seam = w.out_first_pos != nullptr;
    int32_t lhs = 0, rhs = 0; bool want_flank = true;                     // flank sizes in window coordinates (pair_hmm.hpp:572-587)
    if (seam) {
        want_flank = w.seam_lhs != nullptr;
        if (want_flank && active) { lhs = w.seam_lhs[ti]; rhs = w.seam_rhs[ti]; }
    } else if (active) {
        const uint32_t g = w.hap_region[t.hap];
        lhs = (int32_t)w.reg_lhs[g];
        if (lhs < off) lhs = 0; else { lhs -= off; if (lhs < 0) lhs = 0; }
        rhs = (int32_t)w.reg_rhs[g];
        if (off + L < Lh - rhs) rhs = 0; else { rhs += off + L; rhs -= Lh; if (rhs < 0) rhs = 0; }
    }
    const int32_t rhs_begin = L - rhs;
    const int32_t n_diag = 2 * (T + B) + 1; const int64_t n_flat = (int64_t)n_diag * B;
    // this task row's lines: tile kt, band lane b at (kt * 64 + b) * 16 words. A row without a task (the tail of the last wave: its group lies behind the scratch of a host-sized launch)
    // points at the scratch's first line: it walks nothing, but the next-window fetch below is issued by every row of the wave
    const uint32_t* bpg = w.bp + (active ? ((size_t)group * w.k_cap * 64 + row * B) * 16 : (size_t)0);
    const uint32_t hshift = 16 * half;

    int32_t sidx = end.sidx, i = sidx / 2 - T, y = T, x = sidx - T;       // walker state (set_alignments :180-193), the same in all sixteen lanes of the row
    int32_t flank = 0, msz = 0; uint32_t nev = 0, state = 0;
    bool ok = active && sidx >= 0, fin = !ok;
    if (ok) { const int64_t f0 = (int64_t)sidx * B + i; if (f0 < 0 || f0 >= n_flat) { ok = false; fin = true; } }   // :186-190

    WalkPricing pricing;
    {
        const bool fwd = !w.rrev[t.read];
        const size_t hb0 = (size_t)ho + (uint32_t)off;
        pricing.rbases = w.rbases + ro; pricing.rquals = (const int8_t*)w.rquals + ro; pricing.hbases = w.hbases + hb0;
        pricing.mask = (fwd ? w.maskF : w.maskR) + hb0; pricing.prior = (fwd ? w.priorF : w.priorR) + hb0;
        pricing.go = w.go + hb0; pricing.ge = w.ge + hb0;
    }
    const int32_t stop_below_x = (w.early_stop && lhs == 0) ? rhs_begin : INT32_MIN;          // see k_walk
    auto in_flank_at = [&](int32_t xx) { return want_flank && (xx < lhs || xx >= rhs_begin); };   // calculate_flank_score_helper :383-424
    auto word_from_memory = [&](int64_t flat) -> uint32_t {                                     // any cell by flat index = diagonal * B + lane
        const int32_t s = (int32_t)(flat / B), li = (int32_t)(flat % B);
        if (s >= 2 * (T + B)) return 0;                                                         // last row of the reference's array is never written
        const uint32_t k = (uint32_t)s >> 1;
        return bpg[((size_t)(k >> 4) * 64 + (uint32_t)li) * 16 + (k & 15)];
    };
    if (ok) {                                                                                   // the first move only reads the end cell's own label (:191-192)
        const uint32_t wv = word_from_memory((int64_t)sidx * B + i);
        state = (wv >> (hshift + 6 * ((uint32_t)sidx & 1u))) & 3u;
        sidx -= 2;
    }
    // one alignment column from the backpointer word of cell (sidx, i): k_walk's step()
    auto step = [&](uint32_t wv) __attribute__((always_inline)) {
        const uint32_t par = (uint32_t)sidx & 1u;
        const uint32_t bits = (wv >> (hshift + 6 * par)) & 63u, mism = (wv >> (hshift + 15 - par)) & 1u;
        const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;                 // :200
        const bool isM = state == 0, isI = state == 1, isD = !isM && !isI;
        i += isI ? (sidx & 1) : 0;                                                              // insert :205-209
        sidx -= isM ? 2 : 1;                                                                    // match :201-204
        i -= isD ? (sidx & 1) : 0;                                                              // delete :210-215
        x -= isI ? 0 : 1; y -= isD ? 0 : 1;
        const bool inf = in_flank_at(x);
        msz += (inf && !isD) ? 1 : 0;
        flank += (inf && isI) ? w.nuc_prior : 0;
        if (inf && (!isM || mism)) {
            const bool ext = isI ? (y != 0 && new_state == 1) : new_state == 3;                 // first alignment column has prev_state = match (:369)
            const int32_t xi = x - 1 < 0 ? 0 : x - 1;                                           // x-1 == -1 is out of bounds in the reference (UB): clamp
            const uint32_t kind = isM ? 0u : (ext ? 2u : 1u); const int32_t ex = isI ? xi : x, ey = isM ? y : ex;
            OCT_WALK_EVENT(kind, ex, ey, nev < kWalkRowEvents, { if (l16 == 0) evbuf[nev] = e_; ++nev; });
        }
        state = new_state;
        if (y <= 0 || x < stop_below_x) fin = true;                                             // :194 / early stop
    };

    const int32_t mid_lo = want_flank ? lhs : INT32_MIN, mid_hi = (want_flank || stop_below_x != INT32_MIN) ? rhs_begin : INT32_MAX;   // mid_lo < x < mid_hi: the next column lies outside both flanks
    int32_t st_top = INT32_MIN / 2;                                       // the staged window: tiles st_top, st_top - 1, ... st_top - K + 1 (nothing yet)
    uint4 pf[K * LPL][4]; int32_t pf_top = INT32_MIN / 2;                 // the window after it, on its way: tiles pf_top ... pf_top - K + 1
#pragma unroll
    for (uint32_t a = 0; a < K * LPL; ++a)
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) pf[a][j] = make_uint4(0, 0, 0, 0);
    for (;;) {
        if (!fin && sidx < 0) { ok = false; fin = true; }                                       // ran off the first diagonal with target bases left (:195-199)
        if (hw::ballot(!fin) == 0) break;
        const int32_t k = sidx >> 1, kt = k >> 4, kk = k & 15, sidx_at_pass_start = sidx;
        const bool in_band = !fin && (uint32_t)i < (uint32_t)B;
        // ---- stage the next K tiles of this task row (lane l: the lines of band lane l, l + 16, ...) ----
        // The window after this one is known (the walk's diagonal only falls: tiles kt - K ... kt - 2K + 1), so its lines are fetched while this one is walked and wait
        // in registers (`pf`): a 13 kb read's walk was ~200 dependent fetches of a window each (round 6; OCT_WALK_PREFETCH=0 builds without).
        const bool need = in_band && (uint32_t)(st_top - kt) >= K;
        if (hw::ballot(need) != 0) {
            const bool hit = kPrefetch && need && pf_top == kt;
            uint4 v[K * LPL][4];
#pragma unroll
            for (uint32_t a = 0; a < K; ++a)
#pragma unroll
                for (uint32_t q = 0; q < LPL; ++q) {
                    const int32_t tile = kt - (int32_t)a; const uint32_t b = l16 + 16 * q;
                    const bool in = need && !hit && tile >= 0 && b < (uint32_t)B;
                    const uint4* src = (const uint4*)(bpg + ((size_t)(in ? tile : 0) * 64 + (in ? b : 0)) * 16);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) v[a * LPL + q][j] = in ? src[j] : (hit ? pf[a * LPL + q][j] : make_uint4(0, 0, 0, 0));
                }
            hw::wave_lds_fence();                                                               // the reads of the window before are done
#pragma unroll
            for (uint32_t a = 0; a < K; ++a)
#pragma unroll
                for (uint32_t q = 0; q < LPL; ++q) {
                    const uint32_t b = l16 + 16 * q;
                    if (need && b < (uint32_t)B) {
                        uint4* dst = (uint4*)(rowt + (a * B + b) * LS);
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) dst[j] = v[a * LPL + q][j];
                    }
                }
            if (need) st_top = kt;
            hw::wave_lds_fence();
            if constexpr (kPrefetch) {
                // every row of the wave fetches here, with nothing selected on the loaded values (a select would make the wave wait for them now): a row that staged, the window
                // after its new one; a row that did not, the window it is still waiting for, again
                const int32_t nt = need ? kt - (int32_t)K : pf_top;
                pf_top = nt;
#pragma unroll
                for (uint32_t a = 0; a < K; ++a)
#pragma unroll
                    for (uint32_t q = 0; q < LPL; ++q) {
                        const int32_t tile = nt - (int32_t)a; const uint32_t b = l16 + 16 * q;
                        const bool in = tile >= 0 && nt <= (int32_t)w.k_cap && b < (uint32_t)B;          // (tiles below 0 are never walked: whatever tile 0 holds will do)
                        const uint4* src = (const uint4*)(bpg + ((size_t)(in ? tile : 0) * 64 + (in ? b : 0)) * 16);
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) pf[a * LPL + q][j] = src[j];
                    }
            }
        }
        // ---- a run of match columns: words kk, kk - 1, ... of the walk's line, as far as their labels say "match" (the turn itself is a plain step) ----
        const uint32_t par = (uint32_t)sidx & 1u;
        const bool runs = in_band && state == 0;
        const uint32_t* ln = rowt + ((uint32_t)(st_top - kt) * B + (uint32_t)i) * LS;           // (only read where in_band: then the window holds tile kt)
        const uint32_t mine = runs ? ln[l16] : 0u;
        const bool mine_in = runs && (int32_t)l16 <= kk;
        const uint32_t turns = (uint32_t)(hw::ballot(mine_in && ((mine >> (hshift + 6 * par)) & 3u) != 0u) >> rowbase) & 0xffffu;
        int32_t n = 0;
        if (runs) {
            n = turns ? kk - (31 - (int32_t)__builtin_clz(turns)) : kk + 1;
            n = n < y - 1 ? n : y - 1;                                                          // (the step that consumes the last read base is a plain step)
            n = n < x ? n : x;                                                                  // (and so is anything left of the window)
            if (stop_below_x != INT32_MIN) { const int32_t room = x - stop_below_x; n = n < room ? n : room; }   // (and the step that ends an early-stopping walk)
            n = n < 0 ? 0 : n;
        }
        const int32_t j = kk - (int32_t)l16;                                                    // this lane's column of the run: after it x - 1 - j, y - 1 - j
        const bool col = mine_in && j < n, col_fl = col && in_flank_at(x - 1 - j);
        const uint32_t fl_cols = (uint32_t)(hw::ballot(col_fl) >> rowbase) & 0xffffu;
        const uint32_t ev_cols = (uint32_t)(hw::ballot(col_fl && ((mine >> (hshift + 15 - par)) & 1u)) >> rowbase) & 0xffffu;
        if (n > 0 && nev + (uint32_t)__builtin_popcount(ev_cols) > kWalkRowEvents) n = 0;       // no room to queue them: one at a time (which prices on overflow)
        // Long reads (round 4, step 8): a run that went through the whole rest of its line goes on through the NEXT staged lines of the same band lane - a HiFi read walks
        // thousands of match columns straight down one band lane, and one pass per 16 of them (~330 instructions at a lone wave's latency) was 10 ms for 7,000 walks of
        // 13,000 columns. Only between the flanks (nothing to count or queue there: the extension is a pure move) and never for an early-stopping walk.
        int32_t n_ext = 0;
        if (K > 1 && w.k_cap > 16) {                                                              // (wave-uniform: reads longer than ~240 bases; region-sized walks of short reads skip the three ballots)
            bool full = runs && turns == 0u && n == kk + 1 && stop_below_x == INT32_MIN;
            int32_t room = 0;
            if (full) {
                const int32_t x1 = x - n, y1 = y - n;                                           // after the first line's columns
                room = y1 - 1 < x1 ? y1 - 1 : x1;                                               // (the last read base, the window's left edge: plain steps)
                if (want_flank) { const int32_t r = x1 - lhs; room = room < r ? room : r; }     // columns x1 - 1 ... x1 - room stay at or above lhs: outside the left flank
                if (x1 > mid_hi) room = 0;                                                      // (still inside the right flank)
                room = room < 0 ? 0 : room;
            }
#pragma unroll
            for (uint32_t a = 1; a < K; ++a) {
                const bool ext = full && room > 0 && kt >= (int32_t)a && (uint32_t)(st_top - kt) + a < K;
                const uint32_t m2 = ext ? rowt[(((uint32_t)(st_top - kt) + a) * B + (uint32_t)i) * LS + l16] : 0u;
                const uint32_t t2 = (uint32_t)(hw::ballot(ext && ((m2 >> (hshift + 6 * par)) & 3u) != 0u) >> rowbase) & 0xffffu;
                int32_t n2 = ext ? (t2 ? 15 - (31 - (int32_t)__builtin_clz(t2)) : 16) : 0;     // words 15, 14, ... down to the one above the first turn
                n2 = n2 < room ? n2 : room;
                n_ext += n2; room -= n2;
                full = ext && n2 == 16;
            }
        }
        if (n > 0) {
            if ((ev_cols >> l16) & 1u)
                evbuf[nev + (uint32_t)__builtin_popcount(ev_cols & ((1u << l16) - 1u))] = (uint32_t)(x - y) << kWalkEventXBits | (uint32_t)(x - 1 - j);   // (kind 0: a match column; inside the band, so the word holds it)
            nev += (uint32_t)__builtin_popcount(ev_cols); msz += __builtin_popcount(fl_cols);
            sidx -= 2 * n; x -= n; y -= n;
        }
        if (n_ext > 0) { sidx -= 2 * n_ext; x -= n_ext; y -= n_ext; }                           // (only ever behind a first line taken whole: n > 0)
        // ---- plain steps, this row on its own, until the next run starts, the staged window ends or the walk does. The loop holds only what nearly every
        //      step needs (the word is in the window, the event - if any - fits the queue) and keeps its flags in one register word (separate bools become
        //      scalar mask arithmetic around every exit); anything else leaves it for ONE general step below ----
        uint32_t lf = (fin ? 2u : 0u) | (n == 0 ? 4u : 0u);                                     // bit 1: the walk is over; bit 2: no run taken, so at least one step (every pass moves)
        // Between the flanks nothing is priced: a step is a pure move. This is where a junk alignment (a wrong
        // candidate position: ~200 columns that zig-zag through the band) spends its time, at one or two waves per SIMD, i.e. at the full latency of every
        // instruction of the chain word -> label -> next cell -> word: the loop holds nothing else, keeps the walker's state as the shift that picks its
        // label (0 match, 2 insert, 4 delete), folds its range tests into unsigned compares and has ONE exit. (Walks that stop early - no left flank -
        // never get here: the loop below takes all their steps.)
        if (!(lf & 2u) && stop_below_x == INT32_MIN) {
            const int32_t lo_tile = st_top - (int32_t)K + 1 < 0 ? 0 : st_top - (int32_t)K + 1;
            const int32_t sidx_lo = lo_tile * 32; const uint32_t sidx_span = (uint32_t)((st_top + 1) * 32 - sidx_lo);       // the window's diagonals (and none below zero)
            const uint32_t x_span = mid_hi > mid_lo ? (uint32_t)mid_hi - (uint32_t)mid_lo - 1u : 0u;                          // mid_lo < x < mid_hi (none where the flanks meet or overlap)
            uint32_t ssh = state == 3 ? 4u : 2u * state, first = (lf & 4u);                                                  // first: the pass's forced step
            for (;;) {
                // (bitwise on 0 / 1 words: `&&` would become nested branches around the load)
                const uint32_t in = ((uint32_t)(sidx - sidx_lo) < sidx_span ? 1u : 0u) & ((uint32_t)i < (uint32_t)B ? 1u : 0u)
                                  & (((uint32_t)x - (uint32_t)mid_lo - 1u) < x_span ? 1u : 0u) & (y > 0 ? 1u : 0u);
                const uint32_t rel = (uint32_t)(st_top - (sidx >> 5));
                const uint32_t cell = rel * B + (uint32_t)i;                                       // (LS = 20 = 16 + 4: two shift-adds instead of a 64-bit multiply-add)
                static_assert(LS == 20, "staged line stride");
                const uint32_t wv = rowt[((cell << 4) + (cell << 2) + (((uint32_t)sidx >> 1) & 15u)) & (0u - in)];
                const uint32_t par1 = (uint32_t)sidx & 1u, bits = (wv >> (hshift + 6 * par1)) & 63u;
                if ((in ^ 1u) | ((ssh | (bits & 3u) | first) == 0u ? 1u : 0u)) break;               // the window / the band / the flank-free stretch / the read ends, or a run of match columns starts here
                const uint32_t ns = (bits >> ssh) & 3u;                                         // :200
                const uint32_t isM = ssh == 0u ? 1u : 0u, isI = ssh == 2u ? 1u : 0u, isD = ssh == 4u ? 1u : 0u;
                i += (int32_t)(isI & par1) - (int32_t)(isD & (par1 ^ 1u));                      // insert :205-209 / delete :210-215 (after its sidx -= 1 the parity has flipped)
                sidx -= 1 + (int32_t)isM;                                                       // match :201-204
                x -= (int32_t)(isI ^ 1u); y -= (int32_t)(isD ^ 1u);
                ssh = ns == 3u ? 4u : 2u * ns; first = 0u;
            }
            state = ssh == 4u ? 3u : ssh >> 1;
            lf = (y <= 0 ? 2u : 0u) | first;
        }
        for (;;) {
            // the columns inside a flank (written with selects: every exit of a loop costs scalar mask bookkeeping at the same full latency)
            const uint32_t rel = (uint32_t)(st_top - (sidx >> 5));
            const bool can = !(lf & 2u) && sidx >= 0 && (uint32_t)i < (uint32_t)B && rel < K && (x <= mid_lo || x >= mid_hi || stop_below_x != INT32_MIN);   // (an early-stopping walk takes all its steps here)
            const uint32_t wv = rowt[can ? (rel * B + (uint32_t)i) * LS + (((uint32_t)sidx >> 1) & 15u) : 0u];
            const uint32_t par1 = (uint32_t)sidx & 1u;
            const uint32_t bits = (wv >> (hshift + 6 * par1)) & 63u, mism = (wv >> (hshift + 15 - par1)) & 1u;
            const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;             // :200
            const bool isM = state == 0, isI = state == 1, isD = !isM && !isI;
            const int32_t ni = i + (isI ? (int32_t)par1 : 0) - (isD ? (int32_t)(par1 ^ 1u) : 0);   // insert :205-209 / delete :210-215 (after its sidx -= 1 the parity has flipped)
            const int32_t nsidx = sidx - (isM ? 2 : 1);                                         // match :201-204
            const int32_t nx = x - (isI ? 0 : 1), ny = y - (isD ? 0 : 1);
            const bool inf = in_flank_at(nx), ev = inf && (!isM || mism);
            const bool run_here = !(lf & 4u) && isM && (bits & 3u) == 0u;                       // a run of match columns starts here
            if (!can || run_here || (ev && nev >= kWalkRowEvents)) break;                       // (no room in the queue: the general step below prices the event)
            const bool ext = isI ? (ny != 0 && new_state == 1) : new_state == 3;                // first alignment column has prev_state = match (:369)
            const int32_t xi = nx - 1 < 0 ? 0 : nx - 1;                                         // x-1 == -1 is out of bounds in the reference (UB): clamp
            evbuf[ev ? nev : walk_rows_row_words(B) - 1u] =                                             // (all sixteen lanes store the same word; no event: a scratch word behind the queue;
                (isM ? 0u : (ext ? 2u : 1u)) << 30 | (isM ? (uint32_t)(nx - ny) : 0u) << kWalkEventXBits | (uint32_t)(isI ? xi : nx);   //  `can`: the walk is inside the band, so the word holds a match column's diagonal)
            nev += ev ? 1u : 0u;
            msz += (inf && !isD) ? 1 : 0;
            flank += (inf && isI) ? w.nuc_prior : 0;
            i = ni; sidx = nsidx; x = nx; y = ny; state = new_state;
            lf = (ny <= 0 || nx < stop_below_x) ? 2u : 0u;                                      // :194 / early stop
        }
        fin = (lf & 2u) != 0;
        // the general step: outside the band (the flat-index rules), an event to price at once - or a pass that has not moved the walk (whatever the loops' entry
        // conditions left out: every pass takes at least one column, so the kernel ends)
        if (!fin && sidx >= 0 && ((uint32_t)i >= (uint32_t)B || nev >= kWalkRowEvents || sidx == sidx_at_pass_start)) {
            uint32_t wv = 0; bool go = true;
            if (i < 0) { ok = false; fin = true; go = false; }                                  // :195-199
            else if (i >= B) {                                                                  // the reference indexes its array flat: lane overflow reads the next diagonal
                const int64_t f = (int64_t)sidx * B + i;
                if (f >= n_flat) { ok = false; fin = true; go = false; } else wv = word_from_memory(f);
            } else {
                const uint32_t rel = (uint32_t)(st_top - (sidx >> 5));
                if (rel >= K) go = false;                                                       // (the next window first)
                else wv = rowt[(rel * B + (uint32_t)i) * LS + (((uint32_t)sidx >> 1) & 15u)];
            }
            if (go) step(wv);
        }
    }
    const int32_t first_pos = ok ? x : -1;
    hw::wave_lds_fence();
    int32_t part = 0;                                                                           // the queued events, priced by the row's lanes together
    if (ok) for (uint32_t e = l16; e < nev; e += 16) part += walk_price_event(pricing, evbuf[e]);
    for (int m = 1; m < 16; m <<= 1) part += (int32_t)hw::shfl_xor((uint32_t)part, m);
    flank += part;
    if (l16 != 0 || !active) return;
    if (seam) {
        w.out_first_pos[ti] = first_pos;
        if (want_flank) { w.out_flank[ti] = ok ? flank : 0; w.out_mask_size[ti] = ok ? msz : 0; }
        return;
    }
    if (!ok) {
        if (w.pair_key) { w.task_key[ti] = ~0ull; hw::atomic_or_u32(w.err_flags, 1u); }       // simd_align throws HMMOverflow (:811-813)
        return;                                             // populate: lowest(), contributes nothing to the max (pair_hmm.hpp:750-752)
    }
    if (T - msz < 2) flank = 0;                             // :757-759 / :664-665
    const int32_t score = end.score;
    const int32_t pen = flank <= score ? score - flank : flank + score;   // :760-764 / :666-670
    if (w.pair_key) {                                       // align mode: compete for the pair under the reference's tie rules
        const uint32_t p = t.off + (uint32_t)B;
        const uint32_t* P = w.pos + (size_t)t.pair * (uint32_t)w.max_pos; const uint32_t np = w.npos[t.pair];
        unsigned long long order = 0;
        for (uint32_t jj = 0; jj < np; ++jj) if (P[jj] == p) { order = jj + 1; break; }
        const unsigned long long key = (unsigned long long)(uint32_t)pen << 32 | order << 8;
        w.task_key[ti] = key;
        hw::atomic_min_u64(w.pair_key + t.pair, key);
        return;
    }
    hw::atomic_min_i32(w.pair_best + t.pair, pen);
}

// Long reads at bands 128 / 256 (one task per band row, C = B / 64 planes): a batch holds ~10^2 traceback tasks of ~10^4 columns, so the
// lockstep walker above would run two or three waves of mostly idle lanes at a memory round trip per band-lane change. Here ONE WAVE WALKS ONE TASK:
// all 64 lanes carry the same walker state (every branch is wave-uniform, so a step is plain branches instead of selects) and share the memory work -
// lane l keeps the 64-byte line of band lane (base + l) of the current tile in LDS, the lines of the next tile are in flight while this one is
// walked (base = the walk's lane - 32 when they were requested; a walk that drifts out of the 64 staged lanes reads its word from memory), and the
// in-flank events are collected in LDS and priced by all lanes together at the end. Same steps, same events, same result as k_walk.
constexpr uint32_t kWalkLongEvents = 2048;
inline size_t walk_long_lds_bytes() { return (kWalkLongEvents + 2 * 64 * 16) * sizeof(uint32_t); }

template <int B, int C>
OCT_MAX_THREADS(64) OCT_KERNEL(k_walk_long)(WalkParams w)
{
    chain_head_priority<true>();
    static_assert(C > 1 && B == 64 * C, "one task per band row of 64 x C diagonals");
    OCT_DYN_SMEM(smem);
    uint32_t* evbuf = (uint32_t*)smem;                                    // [kWalkLongEvents]
    uint32_t* tbuf = evbuf + kWalkLongEvents;                             // [2][64 lines][16 words]
    const uint32_t lane = hw::thread_idx() & 63u, ti = hw::block_idx();
    const DevTask* tasks = w.tasks; uint32_t n_tasks = w.n_tasks;
    if (w.ref.totals) { uint32_t first; task_list_range(w.ref, first, n_tasks); tasks += first; }
    if (ti >= n_tasks) return;
    const DevTask t = tasks[ti];
    if (t.pair == kPadTask) return;
    const TraceEnd end = w.ends[ti];
    const uint32_t ro = w.roff[t.read]; const int32_t T = (int32_t)(w.roff[t.read + 1] - ro);
    const uint32_t ho = w.hoff[t.hap]; const int32_t Lh = (int32_t)(w.hoff[t.hap + 1] - ho);
    const int32_t L = T + 2 * B - 1, off = (int32_t)t.off;
    const bool seam = w.out_first_pos != nullptr;
    int32_t lhs = 0, rhs = 0; bool want_flank = true;                     // flank sizes in window coordinates (pair_hmm.hpp:572-587)
    if (seam) {
        want_flank = w.seam_lhs != nullptr;
        if (want_flank) { lhs = w.seam_lhs[ti]; rhs = w.seam_rhs[ti]; }
    } else {
        const uint32_t g = w.hap_region[t.hap];
        lhs = (int32_t)w.reg_lhs[g];
        if (lhs < off) lhs = 0; else { lhs -= off; if (lhs < 0) lhs = 0; }
        rhs = (int32_t)w.reg_rhs[g];
        if (off + L < Lh - rhs) rhs = 0; else { rhs += off + L; rhs -= Lh; if (rhs < 0) rhs = 0; }
    }
    const int32_t rhs_begin = L - rhs;
    const int32_t n_diag = 2 * (T + B) + 1; const int64_t n_flat = (int64_t)n_diag * B;
    const uint32_t* bpg = w.bp + (size_t)ti * w.k_cap * C * 1024;        // this task's tiles: [tile][plane i % C][lane i / C][16 iterations]

    int32_t sidx = end.sidx, i = sidx / 2 - T, y = T, x = sidx - T;       // walker state (set_alignments :180-193), the same in every lane
    int32_t flank = 0, msz = 0; uint32_t nev = 0, state = 0;
    bool ok = sidx >= 0, fin = !ok;
    if (ok) { const int64_t f0 = (int64_t)sidx * B + i; if (f0 < 0 || f0 >= n_flat) { ok = false; fin = true; } }   // :186-190

    WalkPricing pricing;
    {
        const bool fwd = !w.rrev[t.read];
        const size_t hb0 = (size_t)ho + (uint32_t)off;
        pricing.rbases = w.rbases + ro; pricing.rquals = (const int8_t*)w.rquals + ro; pricing.hbases = w.hbases + hb0;
        pricing.mask = (fwd ? w.maskF : w.maskR) + hb0; pricing.prior = (fwd ? w.priorF : w.priorR) + hb0;
        pricing.go = w.go + hb0; pricing.ge = w.ge + hb0;
    }
    auto push_event = [&](uint32_t kind, int32_t ex, int32_t ey) {
        OCT_WALK_EVENT(kind, ex, ey, nev < kWalkLongEvents, { if (lane == 0) evbuf[nev] = e_; ++nev; });
    };
    auto in_flank = [&]() { return want_flank && (x < lhs || x >= rhs_begin); };   // calculate_flank_score_helper :383-424
    auto step = [&](uint32_t wv) {                                        // one alignment column from backpointer word `wv` of cell (sidx, i)
        const uint32_t par = (uint32_t)sidx & 1u;
        const uint32_t bits = (wv >> (6 * par)) & 63u, mism = (wv >> (15 - par)) & 1u;
        const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;                 // :200
        if (state == 0) {                                                                       // match :201-204
            sidx -= 2; --x; --y;
            if (in_flank()) { ++msz; if (mism) push_event(0u, x, y); }
        } else if (state == 1) {                                                                // insert :205-209
            i += sidx & 1; sidx -= 1; --y;
            if (in_flank()) { ++msz; flank += w.nuc_prior; const int32_t xi = x - 1 < 0 ? 0 : x - 1; push_event((y != 0 && new_state == 1) ? 2u : 1u, xi, xi); }   // (x - 1 == -1: UB in the reference, clamped)
        } else {                                                                                // delete :210-215
            sidx -= 1; i -= sidx & 1; --x;
            if (in_flank()) push_event(new_state == 3 ? 2u : 1u, x, x);
        }
        state = new_state;
        if (y <= 0) fin = true;                                                                 // :194
    };
    auto word_from_memory = [&](int64_t flat) -> uint32_t {                                     // any cell by flat index = diagonal * B + lane
        const int32_t s = (int32_t)(flat / B), li = (int32_t)(flat % B);
        if (s >= 2 * (T + B)) return 0;                                                         // last row of the reference's array is never written
        const uint32_t k = (uint32_t)s >> 1;
        return bpg[(((size_t)(k >> 4) * C + (uint32_t)li % C) * 64 + (uint32_t)li / C) * 16 + (k & 15)];
    };
    if (ok) {                                                                                   // the first move only reads the end cell's own label (:191-192)
        const uint32_t wv = hw::readfirstlane(word_from_memory((int64_t)sidx * B + i));
        state = (wv >> (6 * ((uint32_t)sidx & 1u))) & 3u;
        sidx -= 2;
    }
    // staging: lane l fetches the line of band lane base + l of tile kt (four 16-byte loads), later parks it in one of the two LDS buffers
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0; int32_t q_base = 0;      // the requested line and the band lane the request started at
    auto request = [&](int32_t kt) __attribute__((always_inline)) {
        q_base = i - 32 < 0 ? 0 : (i - 32 > B - 64 ? B - 64 : i - 32);
        const uint32_t li = (uint32_t)q_base + lane;
        const uint4* l = (const uint4*)(bpg + (((size_t)kt * C + li % C) * 64 + li / C) * 16);
        q0 = l[0]; q1 = l[1]; q2 = l[2]; q3 = l[3];
    };
    auto park = [&](uint32_t buf) __attribute__((always_inline)) {
        uint4* dst = (uint4*)(tbuf + buf * 1024 + lane * 16);
        dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
    };
    // One step per loop iteration (no sweep over iterations that hold no step): the walk's own position names the tile it needs; when that changes - always to the
    // tile below - the lines requested a tile ago are parked and the next request goes out.
    uint32_t cur = 0; int32_t base_cur = 0, kt_cur = (ok && sidx >= 0) ? (sidx >> 1) >> 4 : -1;
    if (kt_cur >= 0) { request(kt_cur); park(0); base_cur = q_base; if (kt_cur > 0) request(kt_cur - 1); }
    hw::wave_lds_fence();
    while (!fin) {
        if (sidx < 0) { ok = false; break; }                                                    // ran off the first diagonal with target bases left (:195-199)
        const int32_t k = sidx >> 1, kt = k >> 4;
        if (kt != kt_cur) {                                                                     // (kt == kt_cur - 1: a step lowers sidx by at most 2)
            hw::wave_lds_fence(); park(cur ^ 1u); base_cur = q_base; cur ^= 1u; kt_cur = kt; hw::wave_lds_fence();
            if (kt > 0) request(kt - 1);                                                        // in flight while this tile is walked
        }
        uint32_t wv;
        const int32_t rel = i - base_cur;
        if (state == 0 && rel >= 0 && rel < 64 && i < B) {
            // A run of match columns walks straight down ONE band lane at ONE diagonal parity: its backpointer words are the words kk, kk - 1, ... 0 of the
            // staged line. Lanes 0 .. kk look at one word each; up to the first whose M label is not "match" (or to the line's end) the steps only count
            // down - sidx by 2, x and y by 1 each - as long as none of them lies inside a flank and the read does not end among them. Long reads spend
            // nearly all of their ~10^4 columns in such runs: the walk then costs a few iterations per tile instead of sixteen.
            const int32_t kk = k & 15;
            const uint32_t par = (uint32_t)sidx & 1u;
            const uint32_t mine = tbuf[cur * 1024 + (uint32_t)rel * 16 + (lane & 15u)];
            const unsigned long long turns = hw::ballot((int32_t)lane <= kk && ((mine >> (6 * par)) & 3u) != 0u) & 0xffffull;
            int32_t n = turns ? kk - (63 - (int32_t)__builtin_clzll(turns)) : kk + 1;               // match steps before the first other label, walking down from kk
            if (want_flank) { const int32_t room_r = x <= rhs_begin ? n : 0, room_l = x - lhs; n = n < room_r ? n : room_r; n = n < room_l ? n : room_l; }   // x - 1 .. x - n outside both flanks
            n = n < y - 1 ? n : y - 1;                                                              // (the step that consumes the last read base takes the plain path)
            if (n > 0) { sidx -= 2 * n; x -= n; y -= n; continue; }
        }
        if (rel >= 0 && rel < 64 && i < B) wv = hw::readfirstlane(tbuf[cur * 1024 + (uint32_t)rel * 16 + ((uint32_t)k & 15u)]);
        else if (i < 0) { ok = false; break; }                                                  // :195-199
        else {
            const int64_t f = (int64_t)sidx * B + i;                                            // i >= B: the reference indexes its array flat (lane overflow reads the next diagonal)
            if (f >= n_flat) { ok = false; break; }
            wv = hw::readfirstlane(word_from_memory(f));                                        // (waited for inside this branch: the lines in flight are not)
        }
        step(wv);               // the same word in every lane, and the compiler knows it: the walker's state lives in scalar registers
    }
    const int32_t first_pos = ok ? x : -1;
    if (ok) {                                                                                   // the queued events, priced by all lanes together
        hw::wave_lds_fence();
        int32_t part = 0;
        const uint32_t n_q = nev < kWalkLongEvents ? nev : kWalkLongEvents;
        for (uint32_t e = lane; e < n_q; e += 64) part += walk_price_event(pricing, evbuf[e]);
        flank += (int32_t)hw::wave_sum_u32((uint32_t)part);
    }
    if (lane != 0) return;
    if (seam) {
        w.out_first_pos[ti] = first_pos;
        if (want_flank) { w.out_flank[ti] = ok ? flank : 0; w.out_mask_size[ti] = ok ? msz : 0; }
        return;
    }
    if (!ok) {
        if (w.pair_key) { w.task_key[ti] = ~0ull; hw::atomic_or_u32(w.err_flags, 1u); }       // simd_align throws HMMOverflow (:811-813)
        return;                                             // populate: lowest(), contributes nothing to the max (pair_hmm.hpp:750-752)
    }
    if (T - msz < 2) flank = 0;                             // :757-759 / :664-665
    const int32_t score = end.score;
    const int32_t pen = flank <= score ? score - flank : flank + score;   // :760-764 / :666-670
    if (w.pair_key) {                                       // align mode: compete for the pair under the reference's tie rules
        const uint32_t p = t.off + (uint32_t)B;
        const uint32_t* P = w.pos + (size_t)t.pair * (uint32_t)w.max_pos; const uint32_t np = w.npos[t.pair];
        unsigned long long order = 0;
        for (uint32_t j = 0; j < np; ++j) if (P[j] == p) { order = j + 1; break; }
        const unsigned long long key = (unsigned long long)(uint32_t)pen << 32 | order << 8;
        w.task_key[ti] = key;
        hw::atomic_min_u64(w.pair_key + t.pair, key);
        return;
    }
    hw::atomic_min_i32(w.pair_best + t.pair, pen);
}

// Align mode, second pass: the task that won its pair walks its backpointers once more (simple per-step walker) and writes the
// alignment as run-length classes of its columns = make_cigar (pair_hmm.hpp:152-188), last column first, plus the mapping position
// target_offset - pad + first_pos (simd_align :817).
template <int B, int TPR, int C>
OCT_KERNEL(k_walk_cigar)(WalkParams w)
{
    constexpr uint32_t ROWS = 64 * C / B, G = TPR * ROWS;
    const uint32_t ti = hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (ti >= w.n_tasks) return;
    const DevTask t = w.tasks[ti];
    if (t.pair == kPadTask) return;
    const unsigned long long key = w.task_key[ti];
    if (key == ~0ull || key != w.pair_key[t.pair]) return;
    const TraceEnd end = w.ends[ti];
    const uint32_t group = ti / G, slot = ti % G, row = slot / TPR, half = slot % TPR;
    const uint32_t ro = w.roff[t.read]; const int32_t T = (int32_t)(w.roff[t.read + 1] - ro);
    const uint8_t* target = w.rbases + ro; const uint8_t* truth = w.hbases + w.hoff[t.hap] + t.off;
    const int32_t n_diag = 2 * (T + B) + 1; const int64_t n_flat = (int64_t)n_diag * B;
    const uint32_t* bpg = w.bp + (size_t)group * w.k_cap * C * 1024;
    auto bits_at = [&](int64_t flat) -> uint32_t {          // 6 backpointer bits of band cell `flat` = diagonal * B + lane
        const int32_t s = (int32_t)(flat / B), i = (int32_t)(flat % B);
        if (s >= 2 * (T + B)) return 0;
        const uint32_t k = (uint32_t)s >> 1;
        const size_t line = C == 1 ? (size_t)(k >> 4) * 64 + row * B + (uint32_t)i : ((size_t)(k >> 4) * C + (uint32_t)i % C) * 64 + (uint32_t)i / C;
        return (bpg[line * 16 + (k & 15)] >> (16 * half + 6 * (s & 1))) & 63u;
    };
    uint32_t* ops = w.cig_ops + (size_t)t.pair * w.cig_cap;
    uint32_t n_ops = 0, cur = 0, run = 0;
    auto column = [&](uint32_t op) {
        if (op == cur) { ++run; return; }
        if (run) { if (n_ops < w.cig_cap) ops[n_ops] = run << 4 | cur; ++n_ops; }
        cur = op; run = 1;
    };
    int32_t sidx = end.sidx, i = sidx / 2 - T, y = T, x = sidx - y;
    uint32_t state = bits_at((int64_t)sidx * B + i) & 3u;   // set_alignments :191 (this task's walk succeeded in k_walk)
    sidx -= 2;
    while (y > 0) {
        const int64_t f = (int64_t)sidx * B + i;
        if (sidx < 0 || i < 0 || f >= n_flat) break;
        const uint32_t bits = bits_at(f);
        const uint32_t new_state = (bits >> (state == 3 ? 4 : 2 * state)) & 3u;
        if (state == 0) { sidx -= 2; --x; --y; column(truth[x] == target[y] ? 7u : 8u); }          // '=' / X
        else if (state == 1) { i += sidx & 1; sidx -= 1; --y; column(1u); }                         // I
        else { sidx -= 1; i -= sidx & 1; --x; column(2u); }                                         // D
        state = new_state;
    }
    if (run) { if (n_ops < w.cig_cap) ops[n_ops] = run << 4 | cur; ++n_ops; }
    w.cig_n[t.pair] = n_ops;
    w.cig_mpos[t.pair] = t.off + (uint32_t)x;               // first_pos = x
}

// ------------------------------------------------------------------------------------------------------------------
// epilogue: penalty -> ln likelihood, mapping-quality mixture, template sum
// ------------------------------------------------------------------------------------------------------------------
// `out` may be the caller-side landing zone itself (pinned host memory, mapped into the device: a one-shot region call has no copy behind its last kernel); then
// host_stats is the run's counter block in pinned memory as well: the first workgroup leaves the sums of the first n_stripes counter stripes in stripe 0, plus
// the error key and the overflow flag where the full copy would have put them.
OCT_KERNEL(k_epilogue)(DevBatch b, double* out, uint64_t out0, uint64_t out1, unsigned long long* host_stats, uint32_t n_stripes)
{
    if (host_stats && hw::block_idx() == 0 && hw::thread_idx() < kStatStride) {
        unsigned long long sum = 0;
        for (uint32_t sl = 0; sl < n_stripes; ++sl) sum += b.stats[(size_t)sl * kStatStride + hw::thread_idx()];
        host_stats[hw::thread_idx()] = sum;
        if (hw::thread_idx() == 0) { host_stats[(size_t)kStatSlots * kStatStride] = *b.err_key; host_stats[(size_t)kStatSlots * kStatStride + 1] = *b.dsl_overflow; }
    }
    const uint64_t o = out0 + (uint64_t)hw::block_idx() * hw::block_dim() + hw::thread_idx();
    const uint64_t o_wave = wave_first_index(out0);
    if (o >= out1) return;
    const uint32_t h = upper_bound_near(b.hap_out_off, b.n_haps + 1, o_wave, o);
    const uint32_t g = b.hap_region[h];
    const uint32_t row = b.reg_row0[g] + (uint32_t)(o - b.hap_out_off[h]);
    const uint32_t r0 = b.row_off ? b.row_off[row] : row, r1 = b.row_off ? b.row_off[row + 1] : row + 1;
    double acc = 0;
    for (uint32_t r = r0; r < r1; ++r) {
        const uint64_t e = b.hap_pair_off[h] + (r - b.reg_read0[g]);
        const uint32_t shared = b.pair_rep ? b.pair_rep[e] : kNoPair;              // k_dedup_verify: this pair's candidates equal another pair's
        const int32_t pen = b.pair_best[shared != kNoPair ? (uint64_t)shared : e];
        const double ln_given_mapped = pen == kNoScore ? kLowest : -kLn10Div10 * (double)pen;
        double res;
        if (b.use_mapq) {                                   // model.cpp:285-300
            int32_t mq = b.rmapq[r];
            if (b.mapq_trigger >= 0 && mq >= b.mapq_trigger) mq = b.mapq_cap & 0xff;
            const double ln_miss = -kLn10Div10 * mq;
            const double ln_mapped = log(1.0 - exp(ln_miss));
            const double x = ln_mapped + ln_given_mapped, y = ln_miss;
            const double lo = x < y ? x : y, hi = x < y ? y : x;      // maths::log_sum_exp, utils/maths.hpp:294-298
            res = hi + log1p(exp(lo - hi));
        } else {
            res = ln_given_mapped;
        }
        res = res > -1e-15 ? 0.0 : res;
        acc = acc + res;
    }
    out[o] = acc;
}

// Align mode: Alignment::likelihood per pair (mapping-quality mixture, model.cpp:416-429) and the alignment itself where the winner
// was an exact match found by the classifier (try_naive_align: the whole read as one '=' run at its position).
OCT_KERNEL(k_epilogue_align)(DevBatch b, uint64_t pair0, uint64_t pair1, double* lik, uint32_t* mpos, uint32_t* n_ops, uint32_t* ops, uint32_t cap)
{
    const uint64_t e = pair0 + (uint64_t)hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (e >= pair1) return;
    const uint32_t h = upper_bound_idx(b.hap_pair_off, b.n_haps + 1, e);
    const uint32_t g = b.hap_region[h];
    const uint32_t r = b.reg_read0[g] + (uint32_t)(e - b.hap_pair_off[h]);
    const unsigned long long key = b.pair_key[e];
    const double ln_given_mapped = key == ~0ull ? kLowest : -kLn10Div10 * (double)(uint32_t)(key >> 32);
    double res = ln_given_mapped;
    if (b.use_mapq) {
        int32_t mq = b.rmapq[r];
        if (b.mapq_trigger >= 0 && mq >= b.mapq_trigger) mq = b.mapq_cap & 0xff;
        const double ln_miss = -kLn10Div10 * mq;
        const double ln_mapped = log(1.0 - exp(ln_miss));
        const double x = ln_mapped + ln_given_mapped, y = ln_miss;
        const double lo = x < y ? x : y, hi = x < y ? y : x;
        res = hi + log1p(exp(lo - hi));
    }
    lik[e] = res > -1e-15 ? 0.0 : res;
    if (key != ~0ull && (key & 1ull)) {
        const uint32_t order = (uint32_t)(key >> 8) & 0xffu;
        mpos[e] = order == 0 ? b.pair_extra[e] : b.pos[e * (uint64_t)b.max_pos + (order - 1)];
        n_ops[e] = 1;
        if (cap) ops[e * (uint64_t)cap] = (b.roff[r + 1] - b.roff[r]) << 4 | 7u;
    }
}

} // namespace octphmm
