/*
 * oct_phmm.h — C ABI of the MI355X pair-HMM haplotype-likelihood engine.
 *
 * This is the drop-in boundary for ONE path of luntergroup/octopus (reference tree at
 * /root/reference, v0.7.4): the batch that HaplotypeLikelihoodArray::populate computes
 * (src/core/models/haplotype_likelihood_array.cpp:51-199) by calling
 * HaplotypeLikelihoodModel::evaluate (src/core/models/haplotype_likelihood_model.cpp:261-320) ->
 * hmm::PairHMM::evaluate (src/core/models/pairhmm/pair_hmm.hpp:831-841) ->
 * simd::PairHMM::align (src/core/models/pairhmm/simd_pair_hmm.hpp:240-324, 438-509).
 *
 * Plain pointers and sizes only. All input pointers are HOST pointers borrowed for the duration of
 * the call; outputs are caller-allocated. The library owns device memory and streams through the
 * handle. One handle per calling thread (the reference copies its model per task,
 * haplotype_likelihood_array.cpp:172); distinct handles may be used concurrently.
 *
 * There is no CPU fallback: without a usable HIP device every compute entry returns
 * OCT_PHMM_ENODEVICE / OCT_PHMM_EHIP.
 */
#ifndef OCT_PHMM_H
#define OCT_PHMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCT_PHMM_ABI_VERSION 1

/* ---- return codes ------------------------------------------------------------------------- */
enum {
    OCT_PHMM_OK               = 0,
    OCT_PHMM_EINVAL           = 1, /* malformed batch (null pointer, non-monotone offsets, a base quality or
                                      penalty outside [0,127], read not contained in haplotype region order) */
    OCT_PHMM_EBAND            = 2, /* simd::PairHMMWrapper::TooLargeBandSizeError, simd_pair_hmm_wrapper.hpp:45-61,229 */
    OCT_PHMM_ESHORT_HAPLOTYPE = 3, /* HaplotypeLikelihoodModel::ShortHaplotypeError, haplotype_likelihood_model.cpp:17-33,244,252 */
    OCT_PHMM_EHIP             = 4, /* HIP runtime error; status->hip_error holds hipError_t */
    OCT_PHMM_ENODEVICE        = 5, /* no gfx950 device visible */
    OCT_PHMM_EUNSUPPORTED     = 6, /* valid request this build does not cover (see DESIGN.md "limits") */
    OCT_PHMM_EOVERFLOW        = 7  /* hmm::HMMOverflow, pair_hmm.hpp:47-64,815-817 (align path only) */
};

/* Sentinel the reference returns for a window that overruns the haplotype or a traceback that
 * overflowed: std::numeric_limits<double>::lowest() (pair_hmm.hpp:736-738,750-752). */
#define OCT_PHMM_LOWEST (-1.7976931348623157e308)

/* ---- configuration: HaplotypeLikelihoodModel::Config (haplotype_likelihood_model.hpp:36-44) -- */
typedef struct oct_phmm_config {
    uint32_t struct_size;                /* sizeof(oct_phmm_config), for ABI growth */
    int32_t  max_indel_error;            /* Config::max_indel_error; band = smallest of {8,16,32,64,128,256} >= this
                                            (simd_pair_hmm_wrapper.hpp:219-241). CLI default 16, Config{} default 8 */
    int32_t  use_int_scores;             /* Config::use_int_scores -> int32 lanes instead of int16 */
    int32_t  use_mapping_quality;        /* Config::use_mapping_quality (default 1) */
    int32_t  mapping_quality_cap;        /* Config::mapping_quality_cap (default 120) */
    int32_t  mapping_quality_cap_trigger;/* Config::mapping_quality_cap_trigger, -1 = none */
    int32_t  use_flank_state;            /* Config::use_flank_state (default 1): if 0 flank states passed in are ignored,
                                            as Caller::compute_haplotype_likelihoods does (caller.cpp:1172) */
    int32_t  nuc_prior;                  /* hmm::Parameters::nuc_prior (pair_hmm.hpp:86), product value 2 */
    int32_t  max_mapping_positions;      /* HaplotypeLikelihoodArray::maxMappingPositions (array.hpp:104) = 10 */
    int32_t  device_id;                  /* HIP device ordinal */
} oct_phmm_config;

/* Fill with the reference's defaults (Config{} + CLI band 16 is NOT applied: max_indel_error = 8). */
void oct_phmm_config_default(oct_phmm_config* cfg);

typedef struct oct_phmm_handle oct_phmm_handle;
struct oct_phmm_error_model;          /* per-haplotype penalty vectors, below */

/* ---- batch description ---------------------------------------------------------------------- */

/* AlignedRead fields the path touches (basics/aligned_read.hpp): sequence, base_qualities,
 * mapping_quality, is_marked_reverse_mapped(), mapped_region().begin(). Reads of one template are
 * consecutive; a likelihood "row" is a template (TemplateMap overload, array.cpp:105) or a single
 * read (ReadMap overload, array.cpp:51). */
typedef struct oct_phmm_reads {
    uint32_t        n_reads;
    const char*     bases;            /* concatenated sequences */
    const uint8_t*  qualities;        /* concatenated phred base qualities, each <= 127 */
    const uint32_t* offsets;          /* [n_reads+1] into bases/qualities */
    const uint8_t*  mapping_quality;  /* [n_reads] */
    const uint8_t*  reverse_strand;   /* [n_reads] AlignedRead::is_marked_reverse_mapped() */
    const int64_t*  ref_begin;        /* [n_reads] mapped_region(read).begin() */
    uint32_t        n_rows;           /* number of output rows */
    const uint32_t* row_offsets;      /* [n_rows+1] first read of each row; NULL => one read per row (n_rows == n_reads) */
} oct_phmm_reads;

/* Haplotype plus the six per-haplotype vectors HaplotypeLikelihoodModel::reset prepares
 * (haplotype_likelihood_model.cpp:60-78): all concatenated with the same offsets. */
typedef struct oct_phmm_haplotypes {
    uint32_t        n_haps;
    const char*     bases;            /* Haplotype::sequence() */
    const uint32_t* offsets;          /* [n_haps+1] */
    const int64_t*  ref_begin;        /* [n_haps] mapped_region(haplotype).begin() */
    const int8_t*   gap_open;         /* haplotype_gap_open_penalities_,  each in [0,127]. All six vector pointers NULL = generate them
                                         from the handle's error model (oct_phmm_set_error_model) */
    const int8_t*   gap_extend;       /* haplotype_gap_extend_penalities_, each in [0,127] */
    const char*     snv_mask_fwd;     /* haplotype_snv_forward_mask_ */
    const int8_t*   snv_prior_fwd;    /* haplotype_snv_forward_priors_, each in [0,127] */
    const char*     snv_mask_rev;     /* haplotype_snv_reverse_mask_ */
    const int8_t*   snv_prior_rev;    /* haplotype_snv_reverse_priors_ */
    const uint8_t*  substitution_mask;/* optional, only read when the six vectors above are NULL (generated by the library): 1 where the haplotype's
                                         own CIGAR against the reference holds a substitution - those bases keep the maximum SNV prior
                                         (repeat_based_snv_error_model.cpp:168-172). NULL = no haplotype carries substitutions. Travels with the call, so
                                         it reaches the region server's batches too. */
} oct_phmm_haplotypes;

/* HaplotypeLikelihoodModel::FlankState (haplotype_likelihood_model.hpp:46-49). */
typedef struct oct_phmm_flank_state {
    uint32_t lhs_flank, rhs_flank;
} oct_phmm_flank_state;

/* Optional: several independent populate() calls (active regions) in one batch. Region g owns rows
 * [row_offsets[g], row_offsets[g+1]) and haplotypes [hap_offsets[g], hap_offsets[g+1]); every row of a
 * region is scored against every haplotype of the same region. NULL => one region. */
typedef struct oct_phmm_regions {
    uint32_t        n_regions;
    const uint32_t* row_offsets;      /* [n_regions+1] */
    const uint32_t* hap_offsets;      /* [n_regions+1] */
    const uint8_t*  has_flank;        /* [n_regions] or NULL (= none) */
    const oct_phmm_flank_state* flank;/* [n_regions] or NULL */
} oct_phmm_regions;

/* Optional precomputed candidate mapping positions (what map_query_to_target emits,
 * utils/kmer_mapper.hpp:120-159), CSR over (haplotype, read-of-its-region) in output order:
 * entry index = read_pair_offset(haplotype) + (read - first read of the region). NULL => the library
 * runs the k-mer mapper on the device. */
typedef struct oct_phmm_positions {
    const uint64_t* offsets;          /* [n_read_pairs+1] */
    const uint32_t* positions;
} oct_phmm_positions;

typedef struct oct_phmm_status {
    int32_t  code;                    /* same as the return value */
    int32_t  hip_error;               /* hipError_t when code == OCT_PHMM_EHIP */
    uint32_t hap_index;               /* OCT_PHMM_ESHORT_HAPLOTYPE: first offending haplotype (iteration order of the serial reference) */
    uint32_t read_index;              /*   ... and read */
    uint32_t required_extension;      /* ShortHaplotypeError::required_extension() */
    char     message[128];
} oct_phmm_status;

/* Work counters of the last run (for GCUPS accounting). */
typedef struct oct_phmm_stats {
    uint64_t n_pairs;                 /* (read, haplotype) pairs = log-likelihoods before template summing */
    uint64_t n_candidates;            /* in-range (read, haplotype, position) evaluations */
    uint64_t n_fast_path;             /* answered by try_naive_evaluate (pair_hmm.hpp:278-319) */
    uint64_t n_dp_score_only;         /* simd align, score only */
    uint64_t n_dp_traceback;          /* simd align + traceback + flank score */
    uint64_t band_cells;              /* sum over DP tasks of 2*B*(T+B) */
    /* of the above, what the library did not run because the pair's candidates (fast-path minimum, task classes, positions, band windows
     * byte for byte) equal those of another pair of the same read, whose result it shares: the reference computes these again */
    uint64_t n_dp_score_only_shared;  /* score-only DP tasks */
    uint64_t n_dp_traceback_shared;   /* traceback DP tasks */
    uint64_t band_cells_shared;       /* their band cells */
    uint64_t n_pairs_shared;          /* pairs that share another pair's result */
} oct_phmm_stats;

/* ---- lifecycle -------------------------------------------------------------------------------- */
/* Number of MI355X (gfx950) devices visible to the process (valid `device_id`s are 0 .. n-1); 0 when there is none. Device-free call. */
int  oct_phmm_device_count(void);
int  oct_phmm_create(const oct_phmm_config* cfg, oct_phmm_handle** out);
void oct_phmm_destroy(oct_phmm_handle* h);
/* HaplotypeLikelihoodModel::pad_requirement() == hmm band size (haplotype_likelihood_model.cpp:55-58). */
int  oct_phmm_band_size(const oct_phmm_handle* h);
const char* oct_phmm_strerror(int code);

/* ---- page-locked host memory for big batches ------------------------------------------------------- */
/* The input arrays and `out` may live anywhere. Where a caller that hands over HUNDREDS of megabytes per call (many regions in one batch) keeps them in
 * page-locked memory, the library skips its staging copies: inputs go to the device straight from the caller's arrays and results land in `out` itself
 * (the DMA engines move ~50 GB/s; a staged copy through host threads ~15). Any page-locked memory the HIP runtime knows will do (hipHostMalloc,
 * hipHostRegister); these two calls provide it without a HIP dependency in the caller. Detected per array, per call, for batches above 8 MB only -
 * region-sized calls (caller.cpp:1159-1196) never pay for the question. NULL on failure. */
void* oct_phmm_host_alloc(size_t bytes);
void  oct_phmm_host_free(void* p);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* HaplotypeLikelihoodArray::populate: out[out_offset(hap) + row - first row of the region] =
 * ln p(row | haplotype); for one region that is an H x n_rows row-major matrix, haplotype-major, i.e.
 * likelihoods_[h][*][row] flattened (array.hpp:123). `out` has sum over regions of H_g * rows_g doubles. */
int oct_phmm_populate(oct_phmm_handle* h,
                      const oct_phmm_reads* reads,
                      const oct_phmm_haplotypes* haps,
                      const oct_phmm_regions* regions,      /* NULL => one region using `flank` */
                      const oct_phmm_flank_state* flank,    /* NULL => no flank state (single-region form only) */
                      const oct_phmm_positions* positions,  /* NULL => device k-mer mapping */
                      double* out,
                      oct_phmm_status* status);

/* Split form of the same call for callers that keep a batch resident in HBM and overlap transfers:
 * upload (H2D + table building), run (enqueue every kernel of the path on the handle's stream, no
 * host synchronisation beyond what sizing the launches needs), download (sync + D2H). bench.py times
 * oct_phmm_batch_run + oct_phmm_batch_wait only. */
typedef struct oct_phmm_batch oct_phmm_batch;
int  oct_phmm_batch_upload(oct_phmm_handle* h, const oct_phmm_reads*, const oct_phmm_haplotypes*,
                           const oct_phmm_regions*, const oct_phmm_flank_state*, const oct_phmm_positions*,
                           oct_phmm_batch** out, oct_phmm_status* status);
int  oct_phmm_batch_run(oct_phmm_handle* h, oct_phmm_batch* b, oct_phmm_status* status);
int  oct_phmm_batch_wait(oct_phmm_handle* h, oct_phmm_batch* b, oct_phmm_status* status);
int  oct_phmm_batch_download(oct_phmm_handle* h, oct_phmm_batch* b, double* out, oct_phmm_status* status);
int  oct_phmm_batch_stats(const oct_phmm_batch* b, oct_phmm_stats* stats);
/* Test seam: the candidate mapping positions the last run used, pair-major in the order of `out` (pair = haplotype row x read):
 * counts[pair] = how many, positions[pair * max_mapping_positions + j] = the j-th (ascending) — what map_query_to_target
 * (utils/kmer_mapper.hpp:120-159) returns when the device mapped, the caller's own CSR otherwise. Entries past counts[pair] are undefined. */
int  oct_phmm_batch_candidate_positions(oct_phmm_handle* h, oct_phmm_batch* b, uint8_t* counts, uint32_t* positions, oct_phmm_status* status);
size_t oct_phmm_batch_out_size(const oct_phmm_batch* b); /* number of doubles `out` must hold */
/* 1 when the batch runs with device-sized launches: one slice whose traceback scratch fits the host-known task bound, so that
 * oct_phmm_batch_run enqueues the whole step without reading the task counts back (region-sized batches); 0 when the step reads them
 * back to size its DP launches (big batches, alignment). Results are identical either way. Diagnostic / test seam. */
int  oct_phmm_batch_device_sized(const oct_phmm_batch* b);
/* Test / A-B switches (the OCT_PHMM_* names of INTEGRATION.md section 7): a process-wide table consulted when a handle is created or a batch is uploaded.
 * value NULL removes the entry. The library reads such switches from the ENVIRONMENT only when OCT_PHMM_ENV_SWITCHES is set there (the test suite and the
 * tools set it): a caller's environment does not steer the product by accident. None is needed in production. */
int  oct_phmm_test_set(const char* name, const char* value);
/* Diagnostic: the shader clock (GHz) the device sustains over `window_ms` milliseconds, measured by a one-wave kernel on a stream of its own that
 * compares the shader cycle counter with the constant reference clock - i.e. while whatever else is enqueued on the device runs (bench.py prices the
 * VALU roofline of the DP kernels at the clock they actually get). Blocks for the window. */
int  oct_phmm_probe_clock(oct_phmm_handle* h, double window_ms, double* shader_ghz);
/* Average device time (ms) of the last run's dominant DP kernel launches measured with HIP events on the
 * handle's stream, and the number of launches; for bench.py's roofline block. */
int  oct_phmm_batch_kernel_time(const oct_phmm_batch* b, double* dp_kernel_ms, uint32_t* dp_launches);
/* DP launches are bracketed with HIP events only while timing is enabled (off by default: four event records per
 * region-sized call are measurable; also enabled by the environment variable OCT_PHMM_TIMING). */
int  oct_phmm_set_timing(oct_phmm_handle* h, int enabled);
/* The same, split by DP kernel kind: [0] score-only/fast-cost, [1] traceback/fast-cost, [2] score-only/generic,
 * [3] traceback/generic. With more than one slice the launches of different slices overlap on the device, so
 * per-launch durations are only meaningful for a single-slice run (OCT_PHMM_SLICES=1). */
int  oct_phmm_batch_kernel_time_by_kind(const oct_phmm_batch* b, double dp_kernel_ms[4], uint32_t dp_launches[4]);
void oct_phmm_batch_free(oct_phmm_handle* h, oct_phmm_batch* b);

/* ---- region server: many calling threads, one device queue ---------------------------------------- */
/* Octopus calls populate once per active region from each of its region-task threads (caller.cpp:475, octopus.cpp:867). A
 * region-sized call is bound by its chain of dependent kernels and the runtime serialises submissions per device, so N threads with
 * N handles do not give N times the throughput. A server owns the device's handles and worker threads: calls that
 * arrive while the device is busy are concatenated into one multi-region batch (their regions stay independent: own rows, own
 * haplotypes, own flank state), answered together and scattered back. Results are those of oct_phmm_populate for each call; an
 * error in one region (e.g. OCT_PHMM_ESHORT_HAPLOTYPE) is reported to its caller only. Thread-safe; blocks until the caller's
 * result is in `out`. Calls that bring their own candidate positions are served one by one. A call's arrays are read on the caller's own thread (pointer and
 * offset checks, the scan for non-ACGT bases) and by the worker that packs them; they must stay valid until the call returns. */
typedef struct oct_phmm_server oct_phmm_server;
int  oct_phmm_server_create(const oct_phmm_config* cfg, uint32_t max_regions_per_batch /* 0 = 256 */, oct_phmm_server** out);
/* The same server in front of several GPUs of the node (BASELINE configs[3]: independent active regions sharded over the devices, no
 * collective): two workers per device - each a gatherer and a finisher thread around two handles -, all draining the one queue, so
 * whichever device is free takes the calls that have arrived (work sharing rather than a fixed region -> device map; results do not depend on the device).
 * `cfg->device_id` is ignored. */
int  oct_phmm_server_create_multi(const oct_phmm_config* cfg, const int32_t* device_ids, uint32_t n_devices,
                                  uint32_t max_regions_per_batch /* 0 = 256 */, oct_phmm_server** out);
void oct_phmm_server_destroy(oct_phmm_server* s);
int  oct_phmm_server_populate(oct_phmm_server* s, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
                              const oct_phmm_flank_state* flank, const oct_phmm_positions* positions,
                              double* out, oct_phmm_status* status);
/* oct_phmm_set_error_model for every handle of the server: callers may then leave their six penalty-vector pointers NULL. The model is handed to each
 * worker thread, which installs it on its own handle before the next device batch it starts: a batch that is running keeps the model it started with,
 * calls that are enqueued after this returns are answered with the new one. */
int  oct_phmm_server_set_error_model(oct_phmm_server* s, const struct oct_phmm_error_model* model);
/* calls answered and device batches run so far (calls / batches = achieved batching) */
int  oct_phmm_server_stats(const oct_phmm_server* s, uint64_t* n_calls, uint64_t* n_batches);
/* calls answered per device, in the order of `device_ids` */
int  oct_phmm_server_device_calls(const oct_phmm_server* s, uint64_t* calls_by_device, uint32_t n_devices);

/* ---- genotype read-out on the resident matrix ---------------------------------- */
/* ConstantMixtureGenotypeLikelihoodModel::evaluate (core/models/genotype/constant_mixture_genotype_likelihood_model.cpp:29-330)
 * for whole genotype vectors (the `evaluate(genotypes, model)` helper, constant_mixture_genotype_likelihood_model.hpp:58-75)
 * on the likelihood matrix the last run left in HBM, so that only one double per genotype returns to the host:
 *   ln p(reads | genotype) = sum over primed rows [ ln sum_{h in genotype} p(row | h) - ln ploidy ].
 * One "set" = one such call: genotypes of one ploidy over haplotypes of ONE region, evaluated over the rows
 * [row_begin, row_end) of that region (HaplotypeLikelihoodArray::prime(sample): a sample's rows are contiguous).
 * A genotype is `ploidy` batch haplotype indices in non-decreasing order (Genotype<> keeps its haplotypes sorted,
 * which the reference's zygosity case analysis relies on). ploidy 1 also gives haplotype_filter.cpp's LikelihoodSum. */
#define OCT_PHMM_MAX_PLOIDY 16
typedef struct oct_phmm_genotype_sets {
    uint32_t        n_sets;
    const uint32_t* ploidy;           /* [n_sets] 1..OCT_PHMM_MAX_PLOIDY */
    const uint32_t* gt_offsets;       /* [n_sets + 1] genotypes [gt_offsets[s], gt_offsets[s+1]) belong to set s; gt_offsets[0] == 0 */
    const uint32_t* hap_indices;      /* concatenated over genotypes: ploidy[s] haplotype indices each */
    const uint32_t* row_begin;        /* [n_sets] or NULL (= 0) */
    const uint32_t* row_end;          /* [n_sets] or NULL (= all rows of the region) */
} oct_phmm_genotype_sets;
/* out: gt_offsets[n_sets] doubles. The batch must have been run; blocks until the result is on the host. */
int  oct_phmm_batch_genotype_likelihoods(oct_phmm_handle* h, oct_phmm_batch* b, const oct_phmm_genotype_sets* sets,
                                         double* out, oct_phmm_status* status);

/* ---- per-haplotype penalty vectors ------------------------------------------------ */
/* The six vectors HaplotypeLikelihoodModel::reset (haplotype_likelihood_model.cpp:60-78) obtains from its two error models, computed by the
 * library: gap open / extend from RepeatBasedIndelErrorModel::do_set_penalties (core/models/error/repeat_based_indel_error_model.cpp:67-83)
 * with BasicRepeatBasedIndelErrorModel's table look-ups (basic_repeat_based_indel_error_model.cpp:44-103) over
 * tandem::extract_exact_tandem_repeats(sequence, 1, 5) (lib/tandem/tandem.hpp:497-514); SNV masks and prior caps from
 * BasicRepeatBasedSNVErrorModel::do_evaluate (repeat_based_snv_error_model.cpp:144-179). Bit-identical to those classes.
 *
 * A model is its tables, ALREADY EXPANDED the way the reference's constructors expand them (copy the first min(size, N) entries,
 * fill the rest with the last one: basic_repeat_based_indel_error_model.cpp:15-34, repeat_based_snv_error_model.cpp:20-34);
 * oct_phmm_error_model_expand does exactly that for one table. The reference's built-in parameter sets live in
 * error_model_factory.cpp:220-517; oct_phmm_error_model_default fills in its default_model_config (PCR-free, HiSeq-2500). */
#define OCT_PHMM_INDEL_TABLE 50
#define OCT_PHMM_SNV_TABLE   51
typedef struct oct_phmm_error_model {
    int8_t  at_homopolymer_open[OCT_PHMM_INDEL_TABLE], cg_homopolymer_open[OCT_PHMM_INDEL_TABLE];
    int8_t  dinucleotide_open[OCT_PHMM_INDEL_TABLE], trinucleotide_open[OCT_PHMM_INDEL_TABLE];
    int8_t  homopolymer_extend[OCT_PHMM_INDEL_TABLE], dinucleotide_extend[OCT_PHMM_INDEL_TABLE], trinucleotide_extend[OCT_PHMM_INDEL_TABLE];
    int8_t  snv_caps[3][OCT_PHMM_SNV_TABLE];   /* homopolymer, dinucleotide, trinucleotide penalty caps */
    int32_t use_snv_model;                     /* 0: no SNV model (PacBio sequencers, error_model_factory.cpp:480-483): masks = the haplotype itself, priors = 100 (model.cpp:69-73) */
} oct_phmm_error_model;
void oct_phmm_error_model_default(oct_phmm_error_model* model);
/* The reference's CustomRepeatBasedIndelErrorModel (core/models/error/custom_repeat_based_indel_error_model.cpp), which `--sequence-error-model <file>` builds from a model
 * FILE of motif -> penalty rows (make_error_model(path), error_model_factory.cpp:572-590): see "a model read from a file" below (oct_phmm_custom_indel_model_*). */
/* Every parameter set the reference's factory holds (error_model_factory.cpp:220-517), by the names --sequence-error-model takes (option_parser.cpp:571-573;
 * matched like the reference's operator>>: case-insensitive, "PCR-free" also as "PCRF"):
 *   library preparation  PCR, PCR-free, 10X, MDA;  sequencer  HiSeq-2000, HiSeq-2500, HiSeq-4000, X10, NovaSeq, BGISEQ-500, PacBio, PacBioCCS.
 * NULL or "" stands for the default part (PCR-free / HiSeq-2500). The PacBio sequencers have no SNV model (use_snv_model = 0, :480-483).
 * OCT_PHMM_EINVAL: unknown name (the reference's UnknownLibraryPreparation / UnknownSequencer), or a pair the factory has no entry for (10X and MDA on
 * the PacBio sequencers: the reference's map lookup throws there). _by_label parses the option's own "<library>[.<sequencer>]" form (parse_model_config). */
int  oct_phmm_error_model_by_name(const char* library_preparation, const char* sequencer, oct_phmm_error_model* model);
int  oct_phmm_error_model_by_label(const char* label, oct_phmm_error_model* model);
/* dst[0 .. capacity) = src[0 .. min(n, capacity)) then the last source entry repeated; n >= 1 */
void oct_phmm_error_model_expand(int8_t* dst, uint32_t capacity, const int8_t* src, uint32_t n);
/* The vectors of n_haps haplotypes (concatenated like oct_phmm_haplotypes, same offsets) into caller-allocated arrays of offsets[n_haps]
 * entries each. substitution_mask (may be NULL): 1 where the haplotype's own CIGAR against the reference holds a substitution - those
 * bases keep the maximum prior (repeat_based_snv_error_model.cpp:168-172). Host entry: haplotypes are spread over host threads; no
 * device needed. */
int  oct_phmm_penalty_vectors(const oct_phmm_error_model* model, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                              const uint8_t* substitution_mask,
                              int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd, int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev,
                              oct_phmm_status* status);
/* Give the handle an error model (NULL: take it away). From then on oct_phmm_populate / _batch_upload / _align / _server_populate accept an
 * oct_phmm_haplotypes whose SIX vector pointers are all NULL and generate the vectors inside the call (HaplotypeLikelihoodModel::reset
 * for every haplotype): on host threads for region-sized calls, on the device (one haplotype per lane) from a few thousand haplotypes. */
int  oct_phmm_set_error_model(oct_phmm_handle* h, const oct_phmm_error_model* model);
/* ---- a model read from a file: CustomRepeatBasedIndelErrorModel ---------------------------------------------------------------------- */
/* `--sequence-error-model <path>` (option_collation.cpp:1621-1631): make_error_model(path) (error_model_factory.cpp:572-590) reads the file with make_penalty_map
 * (custom_repeat_based_indel_error_model.cpp:105-159) into a CustomRepeatBasedIndelErrorModel and pairs it with the DEFAULT configuration's SNV model (:587).
 * Rows are "<motif>:<p0>,<p1>,..." (gap-open penalties by number of periods), "<motif>+:..." (gap-extension penalties), '#' comments; a repeat's penalty is looked up
 * by its own motif, then by the row of min(period, 10) letters N, then the default (:68-103). _parse accepts and refuses exactly the texts the reference does
 * (OCT_PHMM_EINVAL where it throws "Bad model" / MalformedErrorModelFile: no open row, a row without numbers, an entry that is not [+-]?digits or does not fit int8,
 * a missing ':'), and reproduces its defaults - entry 0 of the first row in the iteration order of the reference's std::unordered_map (:38-40, :53-58), obtained from
 * the same container filled by the same calls (this library and the reference build against the same libstdc++) - and 3 for extension without '+' rows (hpp:28).
 * _create takes rows the caller already holds (e.g. the reference's two maps) with the two defaults stated; has_extend = 0 means "no extension map" (every repeat gets
 * default_extend), has_extend = 1 with n_extend = 0 an empty one. The vectors are those of RepeatBasedIndelErrorModel::do_set_penalties's vector overload
 * (repeat_based_indel_error_model.cpp:67-83) bit for bit (tests/test_error_model.py against the reference's own class compiled in place). Look-ups are keyed by strings:
 * host threads at every batch size (the built-in models' device path does not apply). */
typedef struct oct_phmm_custom_indel_model oct_phmm_custom_indel_model;
typedef struct oct_phmm_motif_penalties { const char* motif; uint32_t motif_len; const int8_t* penalties; uint32_t n_penalties; } oct_phmm_motif_penalties;
int  oct_phmm_custom_indel_model_parse(const char* text, size_t len, oct_phmm_custom_indel_model** out);
int  oct_phmm_custom_indel_model_create(const oct_phmm_motif_penalties* open, uint32_t n_open, int8_t default_open,
                                        const oct_phmm_motif_penalties* extend, uint32_t n_extend, int32_t has_extend, int8_t default_extend,
                                        oct_phmm_custom_indel_model** out);
void oct_phmm_custom_indel_model_destroy(oct_phmm_custom_indel_model* m);
/* what a model holds (any pointer may be NULL) */
int  oct_phmm_custom_indel_model_info(const oct_phmm_custom_indel_model* m, int8_t* default_open, int8_t* default_extend, uint32_t* n_open_rows, uint32_t* n_extend_rows,
                                      int32_t* has_extend);
/* oct_phmm_penalty_vectors with the gap vectors from `indel` and the SNV vectors from `snv` (NULL: oct_phmm_error_model_default, what the reference pairs a file with) */
int  oct_phmm_custom_penalty_vectors(const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                                     const uint8_t* substitution_mask,
                                     int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd, int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev,
                                     oct_phmm_status* status);
/* oct_phmm_set_error_model / oct_phmm_server_set_error_model with such a model: calls that leave their six vector pointers NULL get them from it (the handle / server keeps
 * its own reference: the caller may destroy `indel` afterwards). A penalty outside [0, 127] in a row then fails the upload like a caller's own vector would.
 * oct_phmm_set_error_model (also with NULL) takes the custom model away again. */
int  oct_phmm_set_custom_error_model(oct_phmm_handle* h, const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv);
int  oct_phmm_server_set_custom_error_model(oct_phmm_server* s, const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv);
/* Test seam: the vectors the last upload of this batch generated (device or host path), concatenated like the haplotypes. */
int  oct_phmm_batch_penalty_vectors(oct_phmm_handle* h, oct_phmm_batch* b, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                    int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status);

/* ---- realignment: best alignment per (read, haplotype) pair ---------------------- */
/* HaplotypeLikelihoodModel::align (haplotype_likelihood_model.cpp:322-431; compute_optimal_alignment :335-395,
 * hmm::align pair_hmm.hpp:861-872, try_naive_align :321-341, simd_align :788-823, make_cigar :152-188) for every
 * (haplotype, read) pair of the batch - what read_realigner.cpp:114-139 calls per read. Same inputs as
 * oct_phmm_populate (row_offsets must be NULL: alignments are per read). Pairs are ordered haplotype-major, reads
 * of the haplotype's region in batch order: pair = sum over earlier haplotypes of their region's reads + read index.
 * CIGAR operations use the BAM encoding length << 4 | op with op I = 1, D = 2, '=' = 7, X = 8
 * (CigarOperation::Flag insertion, deletion, sequenceMatch, substitution). */
#define OCT_PHMM_CIGAR_INS 1u
#define OCT_PHMM_CIGAR_DEL 2u
#define OCT_PHMM_CIGAR_EQ  7u
#define OCT_PHMM_CIGAR_X   8u
typedef struct oct_phmm_alignments {
    uint32_t  max_cigar_ops;          /* capacity of `cigar` per pair; too small -> OCT_PHMM_EINVAL, status.required_extension = needed */
    uint32_t* mapping_position;       /* [n_pairs] Alignment::mapping_position (haplotype offset of the first aligned base) */
    double*   likelihood;             /* [n_pairs] Alignment::likelihood (after the mapping-quality mixture) */
    uint32_t* n_cigar_ops;            /* [n_pairs] */
    uint32_t* cigar;                  /* [n_pairs * max_cigar_ops] */
} oct_phmm_alignments;
/* OCT_PHMM_EOVERFLOW <-> hmm::HMMOverflow (a traceback that left the band, pair_hmm.hpp:811-813);
 * OCT_PHMM_ESHORT_HAPLOTYPE as for populate. */
int  oct_phmm_align(oct_phmm_handle* h, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
                    const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                    const oct_phmm_positions* positions, oct_phmm_alignments* out, oct_phmm_status* status);

/* The reference's realigner maps every read WITHOUT a cap on the number of candidate positions (read_realigner.cpp:128,137 call
 * map_query_to_target with the default max_mapping_positions, kmer_mapper.hpp:120-124); this library keeps at most
 * cfg.max_mapping_positions (<= 15) tied diagonals per pair, in ascending order. A read that lies inside a tandem repeat or a
 * homopolymer longer than itself ties on more diagonals than that, and the reference may pick one of the later ones. After an
 * oct_phmm_align call whose positions came from the device mapper (positions == NULL): counts[pair] = how many candidates the mapper
 * kept for that pair (n_pairs must be the call's pair count), *n_saturated = the pairs with counts == cfg.max_mapping_positions, i.e.
 * whose candidate list may be cut short - the caller re-aligns exactly those with its own uncapped positions (oct_phmm_positions) or its
 * CPU path (INTEGRATION.md section 3b does the latter). Either output may be NULL. */
int  oct_phmm_align_candidate_counts(const oct_phmm_handle* h, uint8_t* counts, size_t n_pairs, uint32_t* n_saturated);

/* ---- test seam: the raw band kernel ------------------------------------------------------------ */
/* simd::PairHMM::align on explicit windows (simd_pair_hmm.hpp:438-509): what the reference's golden tests
 * drive (test/unit/core/models/pair_hmm_tests.cpp:63-85). Window i has truth_len = target_len + 2B - 1.
 * snv_mask == NULL selects the no-mask overload; gap_extend == NULL uses gap_extend_scalar.
 * traceback != 0 also returns first_pos and the two gapped strings (capacity 2*(T+B)+1 each, NUL terminated)
 * and, if lhs_flank != NULL, calculate_flank_score (simd_pair_hmm.hpp:511-549) into flank_score/mask_size. */
int oct_phmm_align_windows(oct_phmm_handle* h, uint32_t n,
                           const char* truth, const uint32_t* truth_offsets,
                           const char* target, const uint8_t* qualities, const uint32_t* target_offsets,
                           const int8_t* gap_open, const int8_t* gap_extend, int32_t gap_extend_scalar,
                           const char* snv_mask, const int8_t* snv_prior,
                           int32_t nuc_prior, int32_t traceback,
                           int32_t* scores, int32_t* first_pos,
                           char* align1, char* align2, const uint32_t* align_offsets,
                           const int32_t* lhs_flank, const int32_t* rhs_flank,
                           int32_t* flank_score, int32_t* target_mask_size,
                           oct_phmm_status* status);

#ifdef __cplusplus
}
#endif
#endif /* OCT_PHMM_H */
