// Device-side data model shared by the kernels (phmm_kernels.hpp) and the host API (oct_phmm.hip).
// gfx950 only: wave64, DPP wave/row shifts, packed-int16 VALU.
#pragma once
#include <stdint.h>
#include "phmm_hw.hpp"

namespace octphmm {

constexpr int      kMaxSlots   = 16;           // candidate slots per (read, haplotype) pair: mapped positions + 1
constexpr int32_t  kNoScore    = 0x7fffffff;   // "lowest()" in the integer penalty domain
constexpr uint32_t kPadTask    = 0xffffffffu;  // DevTask::pair of a padding task
constexpr int      kBlockWaves = 4;            // waves per DP workgroup
constexpr int      kGroupsPerWave = 4;         // task groups each wave works through per workgroup
constexpr uint32_t kStatSlots  = 256;          // statistics counters are striped over this many lines
constexpr uint32_t kStatStride = 16;           // counters per stripe (128 bytes)
constexpr uint32_t kScanMeetSlot = 12;         // stripe 0, a counter no statistic uses: where k_scan_finish's two workgroups add up their traceback totals
constexpr uint32_t kNoPair     = 0xffffffffu;  // DevBatch::pair_rep of a pair that is computed itself
constexpr uint32_t kDedupReps  = 48;           // distinct pairs a read remembers per haplotype segment (k_dedup_match)

// Task kinds = DP kernel variants. "fast" = reads are pure ACGT and the haplotype holds only ACGT (and no
// '0' in its masks), so the match cost is one byte-permute out of a per-position cap table; "generic" = any
// bytes, cost evaluated with the reference's equality tests.
enum Kind : int { kScoreFast = 0, kTraceFast = 1, kScoreGen = 2, kTraceGen = 3, kNumKinds = 4 };

struct DevTask {
    uint32_t pair;   // (read, haplotype) pair index in output order, kPadTask for padding
    uint32_t read;
    uint32_t hap;
    uint32_t off;    // alignment_offset = max(0, position - B): first haplotype base of the band window
};

struct TraceEnd {    // what the DP kernel hands to the walk kernel, per traceback task
    int32_t score;   // (minscore - null_score) >> 2, simd_pair_hmm.hpp:323
    int32_t sidx;    // minscoreidx (-1: never updated)
};

struct DevBatch {
    // reads (AlignedRead fields)
    uint32_t n_reads;
    const uint8_t*  rbases;   const uint8_t* rquals;  const uint32_t* roff;
    const uint8_t*  rmapq;    const uint8_t* rrev;    const int64_t*  rbegin;
    uint8_t*        racgt;                            // [n_reads] 1 = only A/C/G/T
    uint32_t*       rrec; uint32_t rrec_stride;       // [n_reads][rrec_stride] read-side DP records of the fast-cost kernels (k_hap_tables), null when unused
    uint32_t*       rrecW;                            // [n_reads][rrec_stride] the same for the multi-wave streaming kernel k_dp_mw (int32 lanes, bands 128 / 256):
                                                      // v_perm selector of the base's cap byte | raw base (0x100 before / '0' after the read) << 8 | quality << 24
    uint32_t n_rows;          const uint32_t* row_off; // may be null (row == read)
    // haplotypes + the six vectors of HaplotypeLikelihoodModel::reset
    uint32_t n_haps;
    const uint8_t*  hbases;   const uint32_t* hoff;   const int64_t* hbegin;
    const int8_t*   go;       const int8_t*  ge;
    const uint8_t*  maskF;    const int8_t*  priorF;  const uint8_t* maskR;  const int8_t* priorR;
    // per-base DP tables, 8 B per base per strand: {caps or raw, go | ge << 16}
    uint2* tabFastF; uint2* tabFastR; uint2* tabGenF; uint2* tabGenR;
    uint32_t* hclean;                                 // [n_haps] 1 = eligible for the fast kernels
    // derived per haplotype / region
    const uint32_t* hap_region; const uint64_t* hap_out_off; const uint64_t* hap_pair_off;
    uint32_t n_regions;
    const uint32_t* reg_row0; const uint32_t* reg_read0; const uint32_t* reg_lhs; const uint32_t* reg_rhs;
    // candidate mapping positions per pair: pos[e * max_pos + j], j < npos[e] (host-provided or written by k_kmer_map)
    uint32_t* pos; uint8_t* npos;
    // what k_kmer_map learnt about the base mismatches between the read and the haplotype along a pair's ONE mapped position pos[e * max_pos] (it compares the
    // two 6-mer hash sequences there anyway; a hash is two bits per base): state << 14 | i1, state 0 = nothing known, 1 = no mismatch, 2 = exactly one, at read
    // position i1, 3 = two or more. Holds for pure-ACGT reads on pure-ACGT haplotypes (k_classify checks); null when another mapper or the caller made the positions.
    uint16_t* pair_mm;
    // 6-mer tables per haplotype (k_kmer_tables): bin32[h * 4096 + hash] = start | occupancy << 16, bin_idx[hoff[h] + slot]
    uint32_t* bin32;                                  // bin32[h * 4096 + hash] = start | occupancy << 16 (k_kmer_map, k_kmer_map_lanes); null when unused
    uint16_t* hhash;                                  // hhash[hoff[h] + p]: 6-mer hash of haplotype h at p (k_kmer_tables); k_kmer_map's exact-count shortcut
    int map_count_only;                               // test / A-B switch: every pair takes the counting path
    int map_stats;                                    // OCT_PHMM_MAP_STATS: count the pairs the shortcut decides (diagnostics)
    // k_kmer_map_lanes' bit-parallel pass: every read's bases as 2-bit kmer_code()s, 16 per dword (base i of a dword in bits 2i, 2i + 1: twelve consecutive bits ARE a 6-mer hash),
    // transposed in tiles of 64 reads - dword j of read r at rcode[((r >> 6) * rcode_words + j) * 64 + (r & 63)], so that a wave's 64 reads load one dword each from 256
    // consecutive bytes; zero behind a read's last base. Null when unused.
    uint32_t* rcode; uint32_t rcode_words;
    uint32_t hash_segs;                               // waves per read in k_kmer_tables' read-hash role: ceil(longest read / 1024 bases), at least 1
    uint16_t* bin_idx; uint16_t* rhash;   // rhash[roff[r] + q]: 6-mer hash of read r at q (written by the trailing workgroups of k_kmer_tables; null where the lane mapper runs: it reads rcode)
    // per pair
    uint64_t  n_pairs;
    int32_t*  pair_best;      // min phred penalty over candidates, kNoScore = none
    uint32_t* pair_cls;       // 2 bits per candidate slot: 0 none, 1 score-only DP, 2 traceback DP
    uint32_t* pair_extra;     // position of the extra slot (original or shifted original)
    uint4*    pair_cnt;       // [n_pairs + 1] per-kind task counts, exclusive-scanned in place
    // configuration
    int band, nuc_prior, max_pos, use_mapq, mapq_cap, mapq_trigger, wide;   // wide = int32 lanes (Config::use_int_scores)
    // HaplotypeLikelihoodModel::align mode (oct_phmm_align): every candidate is aligned with traceback and the best one per pair is kept
    // under key = penalty << 32 | order << 8 | naive, order = 0 for the read's original position (it wins ties, model.cpp:365), 1 + j for
    // mapped position j (the first maximum wins, :355)
    int align_mode; unsigned long long* pair_key;
    // exact de-duplication of pairs (k_window_*, k_dedup_match, k_dedup_verify), null when off: canon[hoff[h] + off] = the first band window of the region
    // with the bytes of haplotype h's window at off (bases and the six vectors); pair_rep[e] = the pair whose result pair e shares, or kNoPair
    uint32_t* canon; uint32_t* pair_rep; uint32_t* pair_hash; uint32_t window_len;            // pair_hash: k_classify's hash of what decides a pair (0 = no DP task)
    uint32_t  dedup_hash_mask;                                                                 // all ones; a test hook narrows both hashes so that unequal windows / pairs collide (OCT_PHMM_DEDUP_HASH_BITS)
    int32_t*  pair_fast;                                                                       // k_classify's fast-path minimum (pair_best before any DP result lands in it)
    uint32_t* dd_hash; uint32_t* dd_hap; uint32_t* dd_n;                                       // [kDedupReps][n_reads] + [n_reads]: a read's table between the slices of its region
    // counters, kStatSlots stripes of kStatStride: [0] candidates [1] fast path [2] score-only DP [3] traceback DP [4] band cells [5] pairs
    // [6] [7] mapper diagnostics (OCT_PHMM_MAP_STATS) [8] score-only / [9] traceback DP tasks not run because their pair shares another pair's result [10] their band cells [11] such pairs
    unsigned long long* stats;
    unsigned long long* err_key;                      // min over failing pairs of (hap << 32 | read); ~0 = none
    unsigned long long* dsl_overflow;                 // device-sized launches: a traceback list is longer than `dsl_trace_cap` tasks (TaskListRef::overflow)
    uint32_t dsl_trace_cap;                           // 0 = no limit (host-sized launches)
};

struct DedupSeg {     // the haplotypes [hap_lo, hap_hi) of one region inside one slice, and the reads of that region in tiles of 64
    uint32_t region, hap_first, hap_lo, hap_hi, read0, n_reads, tile0;   // hap_first: the region's first haplotype (in whatever slice)
    uint32_t resumes, continues;   // the region's earlier haplotypes lie in the previous slice / its later ones in the next: the reads' tables travel (DevBatch::dd_*)
};

// Device-sized launches (region-sized batches): the task counts of a step exist only in device memory (k_hap_bases' padded
// totals: x score-only fast, y traceback fast, z score-only generic, w traceback generic; late-start traceback tasks x fast, y generic). The six task
// lists lie one behind the other in that order in ONE array; a kernel that is handed `totals` finds its list there and the host never reads the counts
// back in the middle of a step. Launch grids come from the host-known bound (pairs x (max_mapping_positions + 1)); surplus workgroups leave at once.
struct TaskListRef {
    const uint4* totals;        // null = host-sized launch: `tasks` / `n_tasks` of the parameter block are the list itself
    const uint4* totals_late;   // may be null (no late-start lists)
    int list;                   // 0..3 = Kind, 4 = late-start fast, 5 = late-start generic
    const unsigned long long* overflow;   // set by the scan when a traceback list outgrew the scratch the host provisioned: every list then reads as empty
                                          // and the host repeats the step with host-sized launches (oct_phmm_batch_wait)
    int join_late;              // `list` is a traceback kind: the flavour's late-start list (which lies right behind it in the array) belongs to this launch too
};

struct DpParams {
    const DevTask* tasks; uint32_t n_tasks;           // n_tasks is a multiple of the group size (device-sized launch: the array all six lists live in, n_tasks unused)
    TaskListRef ref;
    const uint8_t* rbases; const uint8_t* rquals; const uint32_t* roff; const uint8_t* rrev;
    const uint32_t* hoff; const uint2* tabF; const uint2* tabR;
    const uint32_t* rrec; uint32_t rrec_stride;       // per-read record rows (fast-cost kernels): entry j = read position j - band
    const uint32_t* rrecW;                            // the rows of k_dp_mw (DevBatch::rrecW), same stride
    int32_t*  pair_best;                              // score-only kernels: atomicMin target
    uint32_t* bp; TraceEnd* ends;                     // traceback kernels
    uint32_t  k_cap;                                  // 16-iteration backpointer tiles (4 KB each) per task group in the bp scratch
    uint32_t  t_cap;                                  // longest read in the batch
    uint32_t  lh_cap;                                 // longest haplotype in the batch
    uint32_t  nuc4;                                   // packed {nuc_prior << 2, nuc_prior << 2}
    uint32_t  groups_per_block;
    uint32_t  rec_chunk;                              // k_dp / k_dp_pair: iterations' worth of read records a wave keeps in LDS at a time (a multiple of 4); 0 = the whole reads
    // late traceback start (tasks whose window touches only the RIGHT inactive flank): the traceback words are needed, and written, only for the
    // last iterations; everything before runs the score-only recurrence. 0 = off; 1 = permitted (the walk stops early, WalkParams::early_stop): a task group
    // whose tasks ALL start at or behind the left flank's end (off >= reg_lhs: k_classify's class 3, pure geometry) starts late - the groups of a late-start list
    // by construction, and whichever other group happens to qualify.
    int late; const uint32_t* hap_region; const uint32_t* reg_rhs; const uint32_t* reg_lhs;
    // window-paired task lists (k_pair_sort; packed int16 fast-cost kernels with plain adds only): paired_end[hap] = the list index up to which the tasks of haplotype `hap`
    // lie two by two on ONE haplotype window (same offset, same strand) - task 2i and 2i + 1 of the list share it; null = the list is as k_emit wrote it. task0 = list index of tasks[0]
    // (a traceback list that runs in several launches).
    const uint32_t* paired_end; uint32_t task0;
    uint32_t late_from;                               // the launch's first task that belongs to a late-start list (0: the launch IS one; a joined launch: the length of the traceback list proper) -
                                                      // the groups before it skip the question (three dependent loads per group: 4 % of a 10 M-task traceback launch)
};

struct WalkParams {
    const DevTask* tasks; uint32_t n_tasks; const TraceEnd* ends; const uint32_t* bp; uint32_t k_cap; int band;
    TaskListRef ref;                                  // device-sized launch: see DpParams
    const uint8_t* rbases; const uint8_t* rquals; const uint32_t* roff; const uint8_t* rrev;
    const uint8_t* hbases; const uint32_t* hoff; const int8_t* go; const int8_t* ge;
    const uint8_t* maskF; const int8_t* priorF; const uint8_t* maskR; const int8_t* priorR;
    const uint32_t* hap_region; const uint32_t* reg_lhs; const uint32_t* reg_rhs;
    int nuc_prior;
    int32_t* pair_best;                               // populate path: atomicMin of the flank-adjusted penalty
    // test seam outputs (null on the populate path)
    int32_t* out_first_pos; char* out_align1; char* out_align2; const uint32_t* out_align_off;
    const int32_t* seam_lhs; const int32_t* seam_rhs; int32_t* out_flank; int32_t* out_mask_size;
    // align mode (null / 0 otherwise)
    unsigned long long* pair_key; unsigned long long* task_key; const uint32_t* pos; const uint8_t* npos; int max_pos;
    uint32_t* err_flags;                              // bit 0: a traceback left the band (hmm::HMMOverflow)
    uint32_t* cig_ops; uint32_t* cig_n; uint32_t* cig_mpos; uint32_t cig_cap;   // per pair, operations in REVERSE order
    int early_stop;                                   // populate only, no int16 lane can wrap: a walk that has left the right flank of a window without left flank is done
};

} // namespace octphmm
